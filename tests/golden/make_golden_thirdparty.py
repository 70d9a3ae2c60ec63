#!/usr/bin/env python3
"""Golden fixtures G10 / G11 / G12 — the THREE THIRD-PARTY pieces of the hot path this repository can only restate, recorded from the real
binaries by whoever has them (this image has none of cv2 / Ceres / mmcv, and no network: SURVEY.md section 8c):

  G10  cv2.solvePnPRansac(obj, img, K, zeros(8,1), reprojectionError=thr, iterationsCount=30, flags=cv2.SOLVEPNP_EPNP) — and, without a
       threshold, cv2.solvePnP(..., flags=SOLVEPNP_EPNP) — exactly as the reference calls them
       (monorun/ops/least_squares/pnp_uncert_cpu.py:35-42, :54-58), on the candidate sets of 96 synthetic config-2 objects plus the
       4- and 5-candidate cases (OpenCV >= 3.3's early return, oracle/epnp.inc decision (ii))            -> tests/golden/g10_opencv_epnp_ransac.npz
  G11  the reference's own compiled extension, `lib.pnp_uncert` (monorun/ops/least_squares/src/pnp_uncert_cpu.cpp:245-292, Ceres 1.14), on
       the inputs of the committed LM fixtures G7 / G7b: result_val, result_pose, result_tr                -> tests/golden/g11_ceres_pnp_uncert.npz
  G12  mmcv.ops.roi_align(coord_2d, rois, (28, 28), 1.0, 0, 'avg', True) (monorun/models/roi_heads/monorun_roi_head.py:521-523) on the
       reference's own coord_2d map (monorun/datasets/pipelines/loading.py:67-78 + edge padding) and RoIs that hang over every border
                                                                                                           -> tests/golden/g12_mmcv_roi_align.npz

Each fixture is written only if its library imports; what is missing is reported and skipped.  The consuming tests
(tests/test_thirdparty_golden.py) load a fixture when it exists and SKIP BY NAME when it does not.  Inputs are regenerated from this
repository's deterministic generator and committed fixtures, and are stored inside the fixture next to the outputs, so the tests never need
the libraries themselves.

    python tests/golden/make_golden_thirdparty.py [--ext /path/to/MonoRUn/monorun/ops/least_squares] [--out tests/golden]

--ext: the directory of the reference checkout that holds the built cffi module `_ext` (INSTALL.md:46-47 of the reference).
INTEGRATION.md section 8 has the one-paragraph how-to.  Nothing here copies reference code: the script CALLS the libraries.
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def epnp_cases():
    """(tag, obj (n,3) f32, img (n,2) f32, K (3,3) f32, thr float | None) — the arrays cv2 is handed by pnp_uncert_cpu.py:33-58:
    the istd candidates of an object (ascending point order), its camera matrix, its RANSAC threshold."""
    from monorun_amd import synthetic as syn
    from oracle import oracle as orc
    cases = []
    for seed, B, hw, frac in ((1234, 48, 28, 0.15), (4242, 32, 28, 0.4), (99, 16, 10, 0.15)):
        b = syn.make_batch(B=B, hw=hw, seed=seed, outlier_frac=frac)
        x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
        cand = orc.istd_inlier_mask(istd, np.float32(0.6))
        for i in range(B):
            m = cand[i] if cand[i].sum() > 4 else np.ones_like(cand[i])          # pnp_uncert_cpu.py:23-32
            cases.append((f'config2_seed{seed}_{i}', x3d[i][m].copy(), x2d[i][m].copy(), K[0].copy(), float(thr[i])))
    # exactly five / exactly four candidates with a threshold (the early return of OpenCV >= 3.3), and plain solvePnP without one
    rng = np.random.default_rng(7)
    Kc = np.array([[707.0912, 0, 601.8873], [0, 707.0912, 183.1104], [0, 0, 1]], np.float32)
    for n in (5, 5, 5, 4, 4):
        obj = (rng.random((n, 3)) * np.array([3.9, 1.5, 1.6]) - np.array([1.95, 1.5, 0.8])).astype(np.float32)
        yaw, t = rng.uniform(-3, 3), np.array([rng.uniform(-8, 8), rng.uniform(1, 2), rng.uniform(8, 40)])
        R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        pc = obj.astype(np.float64) @ R.T + t
        img = np.stack([Kc[0, 0] * pc[:, 0] / pc[:, 2] + Kc[0, 2], Kc[1, 1] * pc[:, 1] / pc[:, 2] + Kc[1, 2]], 1).astype(np.float32)
        cases.append((f'exactly_{n}_candidates', obj, img, Kc.copy(), 3.0))
    b = syn.make_batch(B=8, hw=10, seed=5)
    x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
    for i in range(8):
        cases.append((f'plain_solvepnp_{i}', x3d[i].copy(), x2d[i].copy(), K[0].copy(), None))
    return cases


def make_g10(out):
    cv2 = importlib.import_module('cv2')
    rec = dict(tag=[], n=[], obj=[], img=[], K=[], thr=[], ok=[], rvec=[], tvec=[], inliers=[])
    for tag, obj, img, K, thr in epnp_cases():
        dist = np.zeros((8, 1), np.float32)                                                     # pnp_uncert_cpu.py:178
        if thr is not None:
            ok, rvec, tvec, inl = cv2.solvePnPRansac(obj, img, K, dist, reprojectionError=thr, iterationsCount=30, flags=cv2.SOLVEPNP_EPNP)
            inl = np.zeros(0, np.int64) if inl is None else np.asarray(inl).reshape(-1).astype(np.int64)
        else:
            ok, rvec, tvec = cv2.solvePnP(obj, img, K, dist, flags=cv2.SOLVEPNP_EPNP)
            inl = np.arange(len(obj), dtype=np.int64)
        m = np.zeros(len(obj), bool)
        m[inl] = True
        rec['tag'].append(tag); rec['n'].append(len(obj)); rec['obj'].append(obj); rec['img'].append(img); rec['K'].append(K)
        rec['thr'].append(np.nan if thr is None else thr); rec['ok'].append(bool(ok))
        rec['rvec'].append(np.asarray(rvec, np.float64).reshape(3) if rvec is not None else np.zeros(3))
        rec['tvec'].append(np.asarray(tvec, np.float64).reshape(3) if tvec is not None else np.zeros(3)); rec['inliers'].append(m)
    np.savez_compressed(os.path.join(out, 'g10_opencv_epnp_ransac.npz'), opencv_version=cv2.__version__, tag=np.array(rec['tag']), n=np.array(rec['n']),
                        obj=np.concatenate(rec['obj']), img=np.concatenate(rec['img']), K=np.stack(rec['K']), thr=np.array(rec['thr'], np.float64),
                        ok=np.array(rec['ok']), rvec=np.stack(rec['rvec']), tvec=np.stack(rec['tvec']), inliers=np.concatenate(rec['inliers']))
    return f'{len(rec["tag"])} cases, OpenCV {cv2.__version__}'


def make_g11(out, ext_dir):
    if not ext_dir:
        raise ImportError('--ext not given (the directory of the reference checkout that holds the built cffi module _ext)')
    sys.path.insert(0, os.path.abspath(ext_dir))
    ext = importlib.import_module('_ext')
    lib, ffi = ext.lib, ext.ffi
    res = dict(fixture=[], index=[], val=[], pose=[], tr=[])
    for name in ('g7_lm_trajectories', 'g7b_lm_rank_deficient_starts'):
        g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
        for i in range(g['x2d'].shape[0]):
            # the marshalling of pnp_uncert_cpu.py:75-106: contiguous float64 copies, clips = [z_min, u_min, u_max, v_min, v_max]
            p2, p3, w = [np.ascontiguousarray(g[k][i], np.float64) for k in ('x2d', 'x3d', 'w')]
            K = np.ascontiguousarray(g['K'][i], np.float64)
            init = np.ascontiguousarray(g['init'][i], np.float64)
            clips = np.array([0.5, g['ur'][i][0], g['ur'][i][1], g['vr'][i][0], g['vr'][i][1]], np.float64)
            val, pose, tr = np.zeros(1, np.int32), np.zeros(4, np.float64), np.zeros(1, np.float64)
            c = lambda a, t='double *': ffi.cast(t, a.ctypes.data)
            lib.pnp_uncert(c(p2), c(p3), c(w), c(K), c(init), c(val, 'int *'), c(pose), ffi.NULL, c(tr), p2.shape[0], c(clips))
            res['fixture'].append(name); res['index'].append(i); res['val'].append(int(val[0])); res['pose'].append(pose.copy()); res['tr'].append(float(tr[0]))
    np.savez_compressed(os.path.join(out, 'g11_ceres_pnp_uncert.npz'), fixture=np.array(res['fixture']), index=np.array(res['index']), val=np.array(res['val'], np.int32),
                        pose=np.stack(res['pose']), tr=np.array(res['tr']))
    return f'{len(res["val"])} solves through the reference\'s own _ext (Ceres)'


def roi_cases():
    """the reference's coord_2d map of a 375 x 1242 image (mgrid, edge-padded to a multiple of 32: 384 x 1248) and RoIs inside, at and over
    every border; rois (n,5) = (batch index, x1, y1, x2, y2)"""
    from monorun_amd.pose_head import gen_coord_2d
    cmap = np.ascontiguousarray(gen_coord_2d(375, 1242).cpu().numpy().astype(np.float32)).reshape(1, 2, 384, 1248)
    rois = np.array([[0, 100.3, 50.2, 180.9, 130.6], [0, 300.0, 20.0, 330.0, 61.0], [0, -6.5, -3.0, 40.0, 60.0], [0, 1200.0, 330.0, 1243.5, 377.0],
                     [0, 10.0, 10.0, 10.0, 10.0], [0, 0.0, 0.0, 1242.0, 375.0], [0, 1236.0, 0.0, 1250.0, 30.0], [0, 600.2, 370.1, 640.7, 390.0],
                     [0, -30.0, 100.0, -2.0, 160.0], [0, 0.4, 0.4, 27.6, 27.6]], np.float32)
    return cmap, rois


def make_g12(out):
    torch = importlib.import_module('torch')
    ops = importlib.import_module('mmcv.ops')
    mmcv = importlib.import_module('mmcv')
    cmap, rois = roi_cases()
    o = ops.roi_align(torch.from_numpy(cmap), torch.from_numpy(rois), (28, 28), 1.0, 0, 'avg', True)
    np.savez_compressed(os.path.join(out, 'g12_mmcv_roi_align.npz'), mmcv_version=mmcv.__version__, rois=rois, out=o.numpy().astype(np.float32), map_shape=np.array(cmap.shape))
    return f'{len(rois)} RoIs, mmcv {mmcv.__version__}'


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('--ext', default=os.environ.get('MONORUN_EXT_DIR'), help='directory holding the reference\'s built cffi module _ext')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    done = 0
    for name, fn in (('G10 (cv2)', lambda: make_g10(a.out)), ('G11 (the reference\'s _ext / Ceres)', lambda: make_g11(a.out, a.ext)), ('G12 (mmcv)', lambda: make_g12(a.out))):
        try:
            print(f'{name}: written — {fn()}')
            done += 1
        except ImportError as e:
            print(f'{name}: SKIPPED — {e}')
    print(f'{done} of 3 fixtures written into {a.out}; run `python -m pytest tests/test_thirdparty_golden.py -rs` next')
    return 0


if __name__ == '__main__':
    sys.exit(main())

"""The third-party pinning kit: consumers of the fixtures G10 / G11 / G12 that tests/golden/make_golden_thirdparty.py
records from REAL binaries — cv2.solvePnPRansac / solvePnP (EPNP), the reference's compiled `lib.pnp_uncert` (Ceres 1.14), mmcv's roi_align.
None of the three libraries exists in this image, so the fixtures are absent here and the three tests SKIP BY NAME, saying which command
produces the missing file; with a fixture present they turn "parity unpinned" (oracle/epnp.inc, oracle/pnp_oracle.c, oracle/oracle.py
headers) into a measured statement.  `test_the_checkers_run_on_loopback_fixtures` feeds the same checkers fixtures
written by the restatement itself — it proves only that the consuming code runs, and says so.

Bars (stated here, to be tightened by whoever first runs them against real data):
  G10  success flags equal; RANSAC inlier masks: identical for >= 80 % of the RANSAC cases, mean IoU >= 0.97 (the basis any
       eigen-solver picks in the degenerate null space of a five-point M^T M flips individual hypotheses' consensus — a build-dependent choice
       inside OpenCV too); where the masks are identical the re-fitted pose agrees to 1e-6 (rvec) / 1e-6 relative (tvec); plain solvePnP
       (no RANSAC) to the same bars; the 5-candidate early return with orc.set_epnp_cv_early_return on OpenCV >= 3.3; the 4-candidate
       case (P3P inside OpenCV, not restated) on success flag and mask only.
  G11  result_val equal; result_pose within 1e-6 (the north-star bar is 1e-4); result_tr within 1e-6 relative — the trust-region radius at
       exit encodes the whole accept / reject history, so this is the pass-by-pass check a Ceres binary allows.
  G12  every output value within 1e-4 of the restatement (float32 sums of coordinates up to 1248), borders included."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
HOWTO = 'python tests/golden/make_golden_thirdparty.py  (INTEGRATION.md section 8)'


def _load(name, needs):
    f = os.path.join(GOLDEN, name)
    if not os.path.exists(f):
        pytest.skip(f'{name} is absent: it can only be recorded where {needs} exists — {HOWTO}')
    return dict(np.load(f, allow_pickle=False))


def _version_tuple(v):
    out = []
    for part in str(v).split('.')[:3]:
        digits = ''.join(ch for ch in part if ch.isdigit())
        out.append(int(digits) if digits else 0)
    return tuple(out)


def check_g10(g):
    """-> dict of statistics; asserts the bars of the module docstring."""
    early = _version_tuple(g['opencv_version']) >= (3, 3, 0)
    off = np.concatenate([[0], np.cumsum(g['n'])])
    n_ransac = same_mask = 0
    ious, worst_r, worst_t = [], 0.0, 0.0
    orc.set_epnp_cv_early_return(early)
    try:
        for i, tag in enumerate(g['tag']):
            obj, img = g['obj'][off[i]:off[i + 1]], g['img'][off[i]:off[i + 1]]
            cvm = g['inliers'][off[i]:off[i + 1]].astype(bool)
            thr = g['thr'][i]
            if np.isnan(thr):                                     # plain cv2.solvePnP (pnp_uncert_cpu.py:54-58)
                rv, tv, _ = orc.epnp(obj, img, g['K'][i])
                ok, m = True, np.ones(len(obj), bool)
            else:
                r = orc.epnp_ransac(obj, img, g['K'][i], float(thr))
                ok, rv, tv, m = r['ok'], r['rvec'], r['tvec'], r['mask']
            assert ok == bool(g['ok'][i]), (tag, 'success flag')
            if not ok:
                continue
            four = len(obj) == 4                                   # P3P inside OpenCV >= 3.3: not restated
            if not np.isnan(thr) and len(obj) > 5:
                n_ransac += 1
                inter, union = (m & cvm).sum(), (m | cvm).sum()
                ious.append(inter / max(union, 1))
                same_mask += int(np.array_equal(m, cvm))
            else:
                assert np.array_equal(m, cvm), (tag, 'every point is an inlier')
            if np.array_equal(m, cvm) and not four:
                dr = np.abs(rv - g['rvec'][i]).max()
                dt = np.abs(tv - g['tvec'][i]).max() / max(1.0, np.abs(g['tvec'][i]).max())
                worst_r, worst_t = max(worst_r, dr), max(worst_t, dt)
    finally:
        orc.set_epnp_cv_early_return(False)
    stats = dict(cases=len(g['tag']), ransac_cases=n_ransac, identical_masks=same_mask, mean_iou=float(np.mean(ious)) if ious else 1.0,
                 worst_rvec=worst_r, worst_tvec_rel=worst_t, opencv=str(g['opencv_version']))
    assert n_ransac == 0 or (same_mask >= 0.8 * n_ransac and stats['mean_iou'] >= 0.97), stats
    assert worst_r <= 1e-6 and worst_t <= 1e-6, stats
    return stats


def check_g11(g):
    src = {name: dict(np.load(os.path.join(GOLDEN, name + '.npz'))) for name in set(g['fixture'].tolist())}
    worst_pose, worst_tr, n = 0.0, 0.0, 0
    for k, (name, i) in enumerate(zip(g['fixture'], g['index'])):
        f = src[str(name)]
        clips = np.array([0.5, f['ur'][i][0], f['ur'][i][1], f['vr'][i][0], f['vr'][i][1]], np.float64)
        r = orc.pnp_uncert(f['x2d'][i].astype(np.float64), f['x3d'][i].astype(np.float64), f['w'][i].astype(np.float64), f['K'][i].astype(np.float64), f['init'][i], clips)
        assert r['val'] == int(g['val'][k]), (name, i, 'result_val')
        if r['val']:
            worst_pose = max(worst_pose, float(np.abs(r['pose'] - g['pose'][k]).max()))
            worst_tr = max(worst_tr, abs(r['tr'] - g['tr'][k]) / max(abs(g['tr'][k]), 1e-300))
            n += 1
    stats = dict(solves=len(g['val']), usable=n, worst_pose=worst_pose, worst_tr_rel=worst_tr)
    assert worst_pose <= 1e-6 and worst_tr <= 1e-6, stats
    return stats


def check_g12(g):
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden_thirdparty import roi_cases
    cmap, rois = roi_cases()
    assert np.array_equal(rois, g['rois']) and tuple(g['map_shape']) == cmap.shape
    out = orc.roi_align_avg(cmap, rois, (28, 28))
    d = float(np.abs(out - g['out']).max())
    assert d <= 1e-4, d
    return dict(rois=len(rois), worst=d)


def test_g10_opencv_epnp_ransac_against_the_restatement():
    print(check_g10(_load('g10_opencv_epnp_ransac.npz', 'cv2 (opencv-python)')))


def test_g11_ceres_pnp_uncert_against_the_restatement():
    print(check_g11(_load('g11_ceres_pnp_uncert.npz', "the reference's built cffi module _ext (Ceres 1.14)")))


def test_g12_mmcv_roi_align_against_the_restatement():
    print(check_g12(_load('g12_mmcv_roi_align.npz', 'mmcv (mmcv.ops.roi_align)')))


def test_the_checkers_run_on_loopback_fixtures(tmp_path):
    """NOT a parity statement: fixtures in the G10 / G11 / G12 format written by the restatement itself go through the same checkers, so that the
    consuming code is exercised here (formats, offsets, bars); a real fixture replaces the producer, nothing else."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden_thirdparty as mk
    rec = dict(tag=[], n=[], obj=[], img=[], K=[], thr=[], ok=[], rvec=[], tvec=[], inliers=[])
    orc.set_epnp_cv_early_return(True)
    try:
        for tag, obj, img, K, thr in mk.epnp_cases()[::4] + mk.epnp_cases()[-13:]:
            if thr is None:
                rv, tv, _ = orc.epnp(obj, img, K); ok, m = True, np.ones(len(obj), bool)
            else:
                r = orc.epnp_ransac(obj, img, K, thr); ok, rv, tv, m = r['ok'], r['rvec'], r['tvec'], r['mask']
            rec['tag'].append(tag); rec['n'].append(len(obj)); rec['obj'].append(obj); rec['img'].append(img); rec['K'].append(K)
            rec['thr'].append(np.nan if thr is None else thr); rec['ok'].append(ok); rec['rvec'].append(rv); rec['tvec'].append(tv); rec['inliers'].append(m)
    finally:
        orc.set_epnp_cv_early_return(False)
    g10 = dict(opencv_version=np.array('4.5.0-loopback'), tag=np.array(rec['tag']), n=np.array(rec['n']), obj=np.concatenate(rec['obj']), img=np.concatenate(rec['img']),
               K=np.stack(rec['K']), thr=np.array(rec['thr'], np.float64), ok=np.array(rec['ok']), rvec=np.stack(rec['rvec']), tvec=np.stack(rec['tvec']),
               inliers=np.concatenate(rec['inliers']))
    s = check_g10(g10)
    assert s['identical_masks'] == s['ransac_cases'] > 0 and s['worst_rvec'] == 0.0
    g7 = dict(np.load(os.path.join(GOLDEN, 'g7_lm_trajectories.npz')))
    idx = np.arange(0, g7['x2d'].shape[0], 4)
    g11 = dict(fixture=np.array(['g7_lm_trajectories'] * len(idx)), index=idx, val=g7['val'][idx], pose=g7['pose'][idx], tr=g7['radius'][idx])
    s = check_g11(g11)
    assert s['solves'] == len(idx)
    cmap, rois = mk.roi_cases()
    g12 = dict(rois=rois, out=orc.roi_align_avg(cmap, rois, (28, 28)).astype(np.float32), map_shape=np.array(cmap.shape))
    assert check_g12(g12)['worst'] == 0.0

"""R5 on the GPU: the reference's own initialiser — cv2.solvePnPRansac(..., iterationsCount=30, flags=SOLVEPNP_EPNP) inside the
per-object driver, /root/reference/monorun/ops/least_squares/pnp_uncert_cpu.py:33-68 — as `mr_epnp_ransac_batched`, against the
CPU restatement (oracle/epnp.inc; OpenCV itself is third-party and absent: "parity unpinned" against a cv2 binary, pinned against
the restatement's known answers in tests/test_oracle_epnp.py).  Bars: RANSAC masks BIT-EXACT, initial pose
<= 1e-9, pose after the LM <= 1e-4 with identical LM iteration counts and exit reasons."""
import numpy as np
import pytest
import torch

from monorun_amd import synthetic as syn

pytestmark = pytest.mark.gpu
INIT_TOL = 1e-9          # the arithmetic is the restatement's operation for operation; what differs is acos / sin / cos (ulps)
POSE_TOL = 1e-4          # north_star: rotation / translation within 1e-4


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _t(dev, a):
    t = torch.from_numpy(np.asarray(a))
    d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev)
    d.copy_(t)
    return d


def _stage_reference(orc, x2d, istd, x3d, K, thr, istd_thres=0.6, max_iters=30):
    """per object: candidates (pnp_uncert_cpu.py:164-168, :23-32) -> the restated solvePnPRansac with a trace of its hypotheses"""
    cand = orc.istd_inlier_mask(istd, np.float32(istd_thres))
    out = []
    for i in range(x2d.shape[0]):
        m = cand[i] if cand[i].sum() > 4 else np.ones_like(cand[i])
        idx = np.nonzero(m)[0]
        Ki = K.reshape(-1, 9)[i if K.reshape(-1, 9).shape[0] > 1 else 0]
        r = orc.epnp_ransac_trace(x3d[i][idx], x2d[i][idx], Ki, float(thr[i]), max_iters=max_iters)
        full = np.zeros_like(m)
        if r['ok']:
            full[idx] = r['mask']
        else:
            full = m.copy()
        r['full_mask'] = full
        r['n'] = len(idx)
        out.append(r)
    return out


def _check_stage(gpu, refs):
    ini, imask, ivalid, diag, hyp = [a.cpu().numpy() for a in gpu]
    worst_init = 0.0
    for i, r in enumerate(refs):
        ev = r['cnt'] >= 0                                # the iterations the sequential loop really ran
        a, c = hyp[i][:len(ev)][ev], r['hyp'][ev]
        both_nan = np.isnan(a) & np.isnan(c)
        assert np.all(both_nan | (np.abs(a - c) <= 1e-12 * np.maximum(1.0, np.abs(c)))), (i, 'hypotheses')
        assert np.array_equal(imask[i].astype(bool), r['full_mask']), (i, 'RANSAC inlier mask')
        assert bool(ivalid[i]) == r['ok'], (i, 'success flag')
        assert int(diag[i, 0]) == r['iters'] and int(diag[i, 2]) == r['n'], (i, 'iterations / candidates')
        if r['ok']:
            assert int(diag[i, 1]) == int(r['mask'].sum()), (i, 'inlier count of the best model')
            ref = np.array([r['rvec'][1], *r['tvec']])
            worst_init = max(worst_init, np.abs(ref - ini[i]).max())
        else:
            assert np.all(ini[i] == 0.0)
    assert worst_init <= INIT_TOL, worst_init
    return worst_init


@pytest.mark.parametrize('planar', [True, False])
def test_ransac_stage_object_by_object(dev, orc, planar):
    """config-2 objects, both layouts (numpy's pairwise vs sequential istd mean decides the candidates): every hypothesis the
    sequential RANSAC evaluated, its iteration count, the inlier mask (bit-exact), the success flag and the initial pose."""
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device
    b = syn.make_batch(B=160, seed=1234)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=planar)
    gpu = epnp_ransac_device(_t(dev, x2d), _t(dev, istd), _t(dev, x3d), _t(dev, K), epnp_istd_thres=0.6, epnp_ransac_thres=_t(dev, thr),
                             with_diag=True, debug_hypotheses=True)
    torch.cuda.synchronize()
    refs = _stage_reference(orc, np.ascontiguousarray(x2d), np.ascontiguousarray(istd), np.ascontiguousarray(x3d), K, thr)
    _check_stage(gpu, refs)
    assert sum(r['ok'] for r in refs) >= 155 and max(r['iters'] for r in refs) > 3      # the adaptive iteration count is exercised


def test_hard_objects_many_iterations_failures_and_tiny_sets(dev, orc):
    """heavy outlier shares (the adaptive count stays high: up to 30 iterations replayed), a threshold so small that no model
    reaches 5 inliers (failure: flag 0, zero pose, mask = the istd candidates), exactly 5 candidates (solved directly, all
    inliers), fewer than 5 (failure), per-object cameras."""
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device
    rng = np.random.default_rng(5)
    b = syn.make_batch(B=48, seed=77)
    x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
    x2d, x3d, thr, istd = x2d.copy(), x3d.copy(), thr.copy(), istd.copy()
    P = x2d.shape[1]
    for i in range(0, 16):                                 # 40 - 70 % gross outliers among the candidates
        bad = rng.random(P) < rng.uniform(0.4, 0.7)
        x3d[i, bad] += rng.normal(0, 0.8, (int(bad.sum()), 3)).astype(np.float32)
    thr[16:20] = 1e-4                                      # nothing fits: RANSAC fails
    for i in range(20, 24):                                # exactly five candidates
        istd[i] = 1e-3; istd[i, rng.choice(P, 5, replace=False)] = 1.0
    Kb = np.repeat(K.reshape(1, 3, 3), 48, 0).copy(); Kb[:, 0, 0] *= np.linspace(0.97, 1.03, 48).astype(np.float32); Kb[:, 1, 2] += np.linspace(-4, 4, 48).astype(np.float32)
    gpu = epnp_ransac_device(_t(dev, x2d), _t(dev, istd), _t(dev, x3d), _t(dev, Kb), epnp_istd_thres=0.6, epnp_ransac_thres=_t(dev, thr),
                             with_diag=True, debug_hypotheses=True)
    torch.cuda.synchronize()
    refs = _stage_reference(orc, x2d, istd, x3d, Kb, thr)
    _check_stage(gpu, refs)
    assert max(r['iters'] for r in refs[:16]) >= 20 and not any(r['ok'] for r in refs[16:20]) and all(r['n'] == 5 and r['ok'] for r in refs[20:24])
    # The hypotheses are solved in two rounds (the first `first_round` for every object, the rest for the objects whose replayed loop
    # still wants iterations): where the split falls changes the work, never a bit of the result — against the restatement and
    # against each other (the default, 8, is what ran above)
    for first in (1, 5, 19, 29, 30):
        g = epnp_ransac_device(_t(dev, x2d), _t(dev, istd), _t(dev, x3d), _t(dev, Kb), epnp_istd_thres=0.6, epnp_ransac_thres=_t(dev, thr),
                               with_diag=True, debug_hypotheses=True, first_round=first)
        torch.cuda.synchronize()
        _check_stage(g, refs)
        assert all(torch.equal(a, c) for a, c in zip(g[:4], gpu[:4])), first
    # four points only (P = 4): what OpenCV returns after its single P3P step — EPnP on the four points, all four inliers (oracle/epnp.inc,
    # version-dependent decision (ii)); the kernel follows the restatement
    x4, w4, X4 = np.ascontiguousarray(x2d[:8, :4]), np.ascontiguousarray(istd[:8, :4]), np.ascontiguousarray(x3d[:8, :4])
    g4 = epnp_ransac_device(_t(dev, x4), _t(dev, w4), _t(dev, X4), _t(dev, Kb[:8]), epnp_istd_thres=0.6,
                            epnp_ransac_thres=_t(dev, thr[:8]), with_diag=True, debug_hypotheses=True)
    torch.cuda.synchronize()
    _check_stage(g4, _stage_reference(orc, x4, w4, X4, Kb[:8], thr[:8]))
    assert bool((g4[1] == 1).all())


@pytest.mark.parametrize('max_iters', [1, 5, 12])
def test_iteration_cap_below_the_default(dev, orc, max_iters):
    """iterationsCount other than the reference's 30 (the C ABI takes 1 .. 30): the cap ends the replayed loop — also when it falls
    inside the first round of hypotheses, on a round boundary, or inside the second round (outlier-heavy objects keep the adaptive
    bound high, so the cap is what stops them)."""
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device
    rng = np.random.default_rng(11)
    b = syn.make_batch(B=40, seed=99)
    x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
    x3d = x3d.copy()
    P = x2d.shape[1]
    for i in range(0, 20):
        bad = rng.random(P) < rng.uniform(0.4, 0.7)
        x3d[i, bad] += rng.normal(0, 0.8, (int(bad.sum()), 3)).astype(np.float32)
    refs = _stage_reference(orc, x2d, istd, x3d, K, thr, max_iters=max_iters)
    for first in (None, 3):
        gpu = epnp_ransac_device(_t(dev, x2d), _t(dev, istd), _t(dev, x3d), _t(dev, K), epnp_istd_thres=0.6, epnp_ransac_thres=_t(dev, thr),
                                 max_iters=max_iters, with_diag=True, debug_hypotheses=True, first_round=first)
        torch.cuda.synchronize()
        _check_stage(gpu, refs)
    assert max(r['iters'] for r in refs) == max_iters


def test_plain_epnp_without_thresholds(dev, orc):
    """epnp_ransac_thres=None: cv2.solvePnP(..., SOLVEPNP_EPNP) on the candidates (pnp_uncert_cpu.py:54-58) — the cooperative EPnP
    alone, on 5 ... 784 points."""
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device
    b = syn.make_batch(B=40, seed=31)
    x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
    istd = istd.copy()
    rng = np.random.default_rng(9)
    for i, k in enumerate((5, 6, 7, 9, 33, 64, 65, 127, 128, 129, 300)):          # k candidates (ragged sets around the 64-lane partials)
        istd[i] = 1e-3; istd[i, rng.choice(x2d.shape[1], k, replace=False)] = 1.0
    ini, imask, ivalid, _, _ = epnp_ransac_device(_t(dev, x2d), _t(dev, istd), _t(dev, x3d), _t(dev, K), epnp_istd_thres=0.6)
    torch.cuda.synchronize()
    ini, imask, ivalid = ini.cpu().numpy(), imask.cpu().numpy().astype(bool), ivalid.cpu().numpy().astype(bool)
    cand = orc.istd_inlier_mask(istd, np.float32(0.6))
    for i in range(40):
        m = cand[i] if cand[i].sum() > 4 else np.ones_like(cand[i])
        rvec, tvec, _ = orc.epnp(x3d[i][m], x2d[i][m], K)
        assert np.array_equal(imask[i], m) and ivalid[i]
        assert np.abs(np.array([rvec[1], *tvec]) - ini[i]).max() <= INIT_TOL, i


def test_end_to_end_config2_1024_objects(dev, orc):
    """pnp_uncert(..., initialiser='epnp') — two launches: EPnP/RANSAC, then the LM + covariance from its result — against the
    reference's flow restated (u2d_pnp_epnp): 1024 config-2 objects, masks bit-exact, identical LM iteration counts and exit
    reasons, pose within 1e-4, covariance within 1e-5 relative."""
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device, pnp_uncert_from_init_device
    from monorun_amd.ops import build_pnp
    b = syn.make_batch(B=1024, seed=1234)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    ref = orc.u2d_pnp_epnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, return_init=True, num_threads=0)
    d = [_t(dev, a) for a in (x2d, istd, x3d, K, ur, vr, thr)]
    ini, imask, ivalid, _, _ = epnp_ransac_device(d[0], d[1], d[2], d[3], epnp_istd_thres=0.6, epnp_ransac_thres=d[6])
    valid, pose, cov, tr, mask, diag = [a.cpu().numpy() for a in pnp_uncert_from_init_device(d[0], d[1], d[2], d[3], d[4], d[5], ini, imask, ivalid, z_min=0.5,
                                                                                           inlier_opt_only=True, with_diag=True)]
    r_ret, r_yaw, r_t, r_cov, r_tr, r_mask, r_diag, r_init = ref
    assert np.array_equal(mask.astype(bool), r_mask) and np.array_equal(valid.astype(bool), r_ret)
    assert np.abs(ini.cpu().numpy() - r_init).max() <= INIT_TOL
    assert np.array_equal(diag[:, 0], r_diag[:, 0]) and np.array_equal(diag[:, 2] % 16, r_diag[:, 2]), 'LM iteration counts / exit reasons'
    ok = r_ret
    dyaw = np.abs(np.angle(np.exp(1j * (pose[:, 0] - r_yaw[:, 0]))))
    assert dyaw[ok].max() <= POSE_TOL and np.abs(pose[:, 1:] - r_t)[ok].max() <= POSE_TOL
    scale = np.abs(r_cov[ok]).reshape(ok.sum(), -1).max(1)[:, None, None]
    assert (np.abs(cov[ok] - r_cov[ok]) / scale).max() <= 1e-5
    assert ok.sum() >= 1000
    # the drop-in module with the option set gives the same tuple
    pnp = build_pnp(dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, forward_exact_hessian=False, initialiser='epnp'))
    ret, yaw, t, pcov, pmask = pnp(*d)
    assert torch.equal(ret.cpu(), torch.from_numpy(valid.astype(bool))) and torch.equal(pmask.cpu(), torch.from_numpy(mask.astype(bool)))
    assert torch.equal(yaw.cpu()[:, 0], torch.from_numpy(pose[:, 0])) and torch.equal(t.cpu(), torch.from_numpy(pose[:, 1:]))


@pytest.mark.parametrize('dtype', ['f32', 'f16'])
def test_56x56_tiles(dev, orc, dtype):
    """the config-5 shape (3136 correspondences per object): fp32 storage needs the workspaces to overlay the records (100 KB
    of records + 49 KB of workspaces do not fit 160 KB of LDS), fp16 storage does not."""
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device, pnp_uncert_from_init_device
    b = syn.make_batch(B=48, hw=56, seed=4321)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    if dtype == 'f16':
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).transpose(0, 2, 1))).to(dev).to(torch.float16).permute(0, 2, 1)
        dx2d, distd, dx3d = tt(x2d), tt(istd), tt(x3d)
        x2d, istd, x3d = [a.float().cpu().numpy() for a in (dx2d, distd, dx3d)]          # the oracle sees the rounded values
    else:
        dx2d, distd, dx3d = _t(dev, x2d), _t(dev, istd), _t(dev, x3d)
    ref = orc.u2d_pnp_epnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, return_init=True, num_threads=0)
    ini, imask, ivalid, _, _ = epnp_ransac_device(dx2d, distd, dx3d, _t(dev, K), epnp_istd_thres=0.6, epnp_ransac_thres=_t(dev, thr))
    valid, pose, cov, tr, mask, diag = [a.cpu().numpy() for a in pnp_uncert_from_init_device(dx2d, distd, dx3d, _t(dev, K), _t(dev, ur), _t(dev, vr), ini, imask,
                                                                                           ivalid, z_min=0.5, inlier_opt_only=True, with_diag=True)]
    r_ret, r_yaw, r_t, r_cov, r_tr, r_mask, r_diag, r_init = ref
    assert np.array_equal(mask.astype(bool), r_mask) and np.array_equal(valid.astype(bool), r_ret)
    assert np.abs(ini.cpu().numpy() - r_init).max() <= INIT_TOL
    assert np.array_equal(diag[:, 0], r_diag[:, 0]) and np.array_equal(diag[:, 2] % 16, r_diag[:, 2])
    ok = r_ret
    dyaw = np.abs(np.angle(np.exp(1j * (pose[:, 0] - r_yaw[:, 0]))))
    assert ok.sum() >= 44 and dyaw[ok].max() <= POSE_TOL and np.abs(pose[:, 1:] - r_t)[ok].max() <= POSE_TOL


def test_empty_batch_and_argument_checks(dev):
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device
    from monorun_amd.ops import pnp_uncert
    z = lambda *s: torch.zeros(*s, device=dev)
    ini, imask, ivalid, _, _ = epnp_ransac_device(z(0, 784, 2), z(0, 784, 2), z(0, 784, 3), z(1, 3, 3), epnp_ransac_thres=z(0))
    assert ini.shape == (0, 4) and imask.shape == (0, 784) and ivalid.shape == (0,)
    ret, yaw, t, cov, mask = pnp_uncert(z(0, 784, 2), z(0, 784, 2), z(0, 784, 3), z(1, 3, 3), z(1, 2), z(1, 2), epnp_ransac_thres=z(0), initialiser='epnp')
    assert ret.shape == (0,) and yaw.shape == (0, 1) and t.shape == (0, 3) and cov.shape == (0, 4, 4) and mask.shape == (0, 784)
    with pytest.raises(ValueError):
        pnp_uncert(z(1, 784, 2), z(1, 784, 2), z(1, 784, 3), z(1, 3, 3), z(1, 2), z(1, 2), initialiser='cv2')
    # the module form passes its keywords through (epnp_first_round: result-neutral)
    from monorun_amd.ops import build_pnp
    b = syn.make_batch(B=24, hw=10, seed=3)
    x = [_t(dev, a) for a in syn.pnp_boundary(b, planar=True)]
    o8 = build_pnp(dict(type='PnPUncert', initialiser='epnp'))(*x)
    o30 = build_pnp(dict(type='PnPUncert', initialiser='epnp', epnp_first_round=30))(*x)
    assert all(torch.equal(p, q) for p, q in zip(o8, o30)) and int(o8[0].sum()) >= 20


def test_adversarial_inputs_terminate_and_agree_with_the_restatement(dev, orc):
    """coincident / planar / collinear objects, NaN and Inf correspondences, overflowing coordinates, zero weights, zero thresholds
    (tests/sweeps/gpu_epnp_fuzz.py has the 400-trial version, profiles/r03_epnp_fuzz.txt): both launches terminate, no non-finite pose is
    reported valid, initialiser success flags and inlier masks equal the restatement's (NaN spectra included: every comparison of the
    eigenvalue ranking is false there, and both sides then read column 0)."""
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device, pnp_uncert_from_init_device
    rng = np.random.default_rng(3)
    for trial in range(40):
        B = int(rng.choice([1, 3, 40])); hw = int(rng.choice([3, 4, 8, 10]))
        b = syn.make_batch(B=B, hw=hw, seed=int(rng.integers(1 << 30)))
        x2d, istd, x3d, K, ur, vr, thr = [np.array(a, copy=True) for a in syn.pnp_boundary(b, planar=bool(rng.integers(2)))]
        P = x2d.shape[1]
        mode, sel = trial % 10, rng.uniform(size=B) < 0.5
        if mode == 0: x3d[sel] = 0.0
        elif mode == 1: x2d[sel, rng.integers(P)] = np.nan
        elif mode == 2: istd[sel] = 0.0
        elif mode == 3: x3d[sel] *= 1e20
        elif mode == 4: x3d[sel, :, 1] = 0.0
        elif mode == 5: x3d[sel] = rng.normal(0, 1, x3d[sel].shape).astype(np.float32)
        elif mode == 6: thr[sel] = 0.0
        elif mode == 7: x2d[sel] = np.inf
        elif mode == 8: x3d[sel, :, 0] = 0.0; x3d[sel, :, 1] = 0.0
        elif mode == 9: x3d[sel] = x3d[sel][:, :1]
        with np.errstate(all='ignore'):
            ref = orc.u2d_pnp_epnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, num_threads=0)
        d = [_t(dev, a) for a in (x2d, istd, x3d, K, ur, vr, thr)]
        ini, im, iv, _, _ = epnp_ransac_device(d[0], d[1], d[2], d[3], epnp_istd_thres=0.6, epnp_ransac_thres=d[6])
        out = pnp_uncert_from_init_device(d[0], d[1], d[2], d[3], d[4], d[5], ini, im, iv, z_min=0.5, inlier_opt_only=True)
        torch.cuda.synchronize()
        valid, pose, mask = out[0].cpu().numpy().astype(bool), out[1].cpu().numpy(), out[4].cpu().numpy().astype(bool)
        assert np.isfinite(pose[valid]).all(), (trial, mode)
        assert np.array_equal(iv.cpu().numpy().astype(bool), ref[6][:, 2] != 8), (trial, mode)
        assert np.array_equal(mask, ref[5]), (trial, mode)


def test_workspace_contract_and_the_library_allocated_workspace(dev):
    """mr_epnp_ransac_batched hands its intermediate results over in a workspace: the caller's (>= mr_epnp_workspace_bytes, 256-byte
    aligned; anything smaller or misaligned is refused, nothing is launched) or, with NULL, a stream-ordered allocation of the library's
    own — the two give bit-identical outputs, and a workspace full of garbage does not leak into the results."""
    import ctypes
    from monorun_amd import _lib
    from monorun_amd.ops.least_squares.pnp_uncert import _strides, _DTYPES
    lib = _lib.load()
    assert lib.mr_epnp_workspace_bytes(0, 784) == 0 and lib.mr_epnp_workspace_bytes(4, 3) == 0
    need = int(lib.mr_epnp_workspace_bytes(64, 784))
    assert need > 64 * 30 * (25 * 4 + 12 * 8) and need % 256 == 0              # at least the samples and R | t of every hypothesis
    assert int(lib.mr_epnp_workspace_bytes(128, 784)) > need
    b = syn.make_batch(B=64, seed=77)
    x2d, istd, x3d, K, ur, vr, thr = [_t(dev, a) for a in syn.pnp_boundary(b, planar=True)]

    def call(work, nbytes):
        ini = torch.full((64, 4), -7.0, device=dev, dtype=torch.float64); im = torch.full((64, 784), 9, device=dev, dtype=torch.uint8)
        iv = torch.full((64,), 9, device=dev, dtype=torch.uint8)
        code = lib.mr_epnp_ransac_batched(x2d.data_ptr(), _strides(x2d), istd.data_ptr(), _strides(istd), x3d.data_ptr(), _strides(x3d), _DTYPES[x2d.dtype],
                                          K.data_ptr(), K.shape[0], thr.data_ptr(), 64, 784, 0.6, 0, 30, ini.data_ptr(), im.data_ptr(), iv.data_ptr(), None, None,
                                          work, nbytes, torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        return code, ini.cpu().numpy(), im.cpu().numpy(), iv.cpu().numpy()
    garbage = torch.randint(0, 256, (need + 512,), device=dev, dtype=torch.uint8)
    base = garbage.data_ptr() + (-garbage.data_ptr()) % 256
    c0, ini0, im0, iv0 = call(base, need)
    c1, ini1, im1, iv1 = call(None, 0)
    assert c0 == 0 and c1 == 0
    assert np.array_equal(ini0, ini1) and np.array_equal(im0, im1) and np.array_equal(iv0, iv1) and iv0.min() in (0, 1) and iv0.max() == 1
    c2, ini2, im2, iv2 = call(base, need - 1)                      # too small
    assert c2 != 0
    assert (ini2 == -7.0).all() and (iv2 == 9).all()               # refused before anything was launched
    c3, *_ = call(base + 8, need)                                  # misaligned
    assert c3 != 0


def test_prepared_reference_flow_launches_in_flight(dev):
    """PnPEpnpLaunch (initialiser + LM, own workspace and outputs) submitted round-robin to PnPPipeline: the results of every launch equal
    the one-call-at-a-time results of the same batch, also when the launches overlap on the device."""
    from monorun_amd import PnPEpnpLaunch, PnPPipeline
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device, pnp_uncert_from_init_device
    batches = [[_t(dev, a) for a in syn.pnp_boundary(syn.make_batch(B=256, seed=500 + i), planar=True)] for i in range(3)]
    refs = []
    for x in batches:
        ini, im, iv, _, _ = epnp_ransac_device(x[0], x[1], x[2], x[3], epnp_istd_thres=0.6, epnp_ransac_thres=x[6])
        refs.append(pnp_uncert_from_init_device(x[0], x[1], x[2], x[3], x[4], x[5], ini, im, iv, z_min=0.5, inlier_opt_only=True))
    torch.cuda.synchronize()
    pipe = PnPPipeline(dev, depth=3, record_events=True)
    ls = [PnPEpnpLaunch(*x[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=x[6], inlier_opt_only=True) for x in batches]
    for rep in range(3):
        evs = [pipe.submit(ls[i], slot=i) for i in range(3)]
    pipe.drain()
    assert all(e is not None for e in evs)
    for l, r in zip(ls, refs):
        assert torch.equal(l.valid, r[0]) and torch.equal(l.pose, r[1]) and torch.equal(l.cov, r[2]) and torch.equal(l.tr, r[3]) and torch.equal(l.mask, r[4])
        assert int(l.valid.sum()) > 200
    empty = PnPEpnpLaunch(*[t[:0] if t.shape[0] == 256 else t for t in batches[0][:6]], epnp_ransac_thres=batches[0][6][:0])
    empty.run()
    # the whole sequence (thirteen launches, no allocation, no host synchronisation inside) can be captured in a HIP graph and replayed
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        ls[0].run(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            ls[0].run()
    for t in (ls[0].valid, ls[0].pose, ls[0].cov, ls[0].mask):
        t.zero_()
    g.replay(); torch.cuda.synchronize()
    r = refs[0]
    assert torch.equal(ls[0].valid, r[0]) and torch.equal(ls[0].pose, r[1]) and torch.equal(ls[0].cov, r[2]) and torch.equal(ls[0].mask, r[4])


def test_grouped_calls_equal_the_calls_one_by_one(dev):
    """mr_epnp_ransac_grouped / PnPEpnpGroupLaunch: every launch of the set carries the objects of up to eight calls (separate input and
    output tensors, shared workspace), each call's LM launch follows — bit-identical to the calls one by one: per-object cameras and a
    shared one, with and without the diag output, groups of 1 to 8, a group in flight next to others, and the argument checks."""
    import ctypes
    from monorun_amd import PnPEpnpLaunch, PnPEpnpGroupLaunch, PnPPipeline, _lib
    lib = _lib.load()
    mk = lambda seed, B=192: [_t(dev, a) for a in syn.pnp_boundary(syn.make_batch(B=B, seed=seed), planar=True)]
    batches = [mk(900 + i) for i in range(7)]
    for x in batches[4:6]:                                # two batches with a camera per object
        x[3] = (x[3].reshape(1, 3, 3).repeat(192, 1, 1) * torch.linspace(0.97, 1.03, 192, device=dev)[:, None, None]).contiguous()
    kw = dict(z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True)
    def solo(x, diag=False, fused=False):
        l = PnPEpnpLaunch(*x[:6], epnp_ransac_thres=x[6], with_diag=diag, fused=fused, **kw); l.run(); torch.cuda.synchronize(); return l
    refs = [solo(x, diag=(i < 2)) for i, x in enumerate(batches)]          # the two entry points one after the other
    def same(l, r):
        return (torch.equal(l.valid, r.valid) and torch.equal(l.pose, r.pose) and torch.equal(l.cov, r.cov) and torch.equal(l.tr, r.tr) and torch.equal(l.mask, r.mask)
                and torch.equal(l.init_pose, r.init_pose) and torch.equal(l.init_mask, r.init_mask) and torch.equal(l.init_valid, r.init_valid))
    # the default of a single launch: the re-fit's pose candidates as the LM launch's prologue (MR_EPNP_DEFER_REFIT + mr_pnp_uncert_from_epnp_grouped)
    for i, x in enumerate(batches):
        l = solo(x, diag=(i < 2), fused=True)
        assert same(l, refs[i]), i
        if i < 2:
            assert torch.equal(l.init_diag, refs[i].init_diag) and torch.equal(l.diag, refs[i].diag)
    for members, diag in (([0, 1], True), ([2, 3, 6], False), ([0, 1, 2, 3], False), ([4, 5], False), ([6], False), ([0, 1, 2, 3, 6], False), ([6, 3, 2, 1, 0, 6, 3, 2], False)):
        for lm in ('fused', 'grouped', 'side_by_side', 'serial'):   # one LM launch over the set (with / without the re-fit in it) / one per member
            ls = [PnPEpnpLaunch(*batches[i][:6], epnp_ransac_thres=batches[i][6], with_diag=diag, **kw) for i in members]
            g = PnPEpnpGroupLaunch(ls, lm=lm)
            g.run(); g.run(); torch.cuda.synchronize()
            for l, i in zip(ls, members):
                assert same(l, refs[i]), (members, i, lm)
                assert int(l.valid.sum()) > 150
                if diag:
                    assert torch.equal(l.init_diag, refs[i].init_diag) and torch.equal(l.diag, refs[i].diag)
    # groups in flight: three groups of two on a depth-3 pipeline, twice over
    ls = [PnPEpnpLaunch(*batches[i][:6], epnp_ransac_thres=batches[i][6], **kw) for i in (0, 1, 2, 3, 6, 0)]
    groups = [PnPEpnpGroupLaunch(ls[2 * k:2 * k + 2]) for k in range(3)]
    pipe = PnPPipeline(dev, depth=3, record_events=False)
    for rep in range(2):
        for k, g in enumerate(groups):
            pipe.submit(g, slot=k)
    pipe.drain()
    for l, i in zip(ls, (0, 1, 2, 3, 6, 0)):
        assert same(l, refs[i])
    # members must agree in shape / camera batching; more than eight calls and mixed thresholds are refused by the library
    with pytest.raises(ValueError):
        PnPEpnpGroupLaunch([refs[0], refs[4]])
    with pytest.raises(ValueError):
        PnPEpnpGroupLaunch([refs[0], solo(mk(77, B=64))])
    with pytest.raises(ValueError):
        PnPEpnpGroupLaunch([refs[0], refs[1], refs[2], refs[3], refs[6]] * 2)
    g = PnPEpnpGroupLaunch([refs[2], refs[3]])
    bad = list(g.args); bad[0] = 9
    assert lib.mr_epnp_ransac_grouped(*bad, None) == -1
    bad = list(g.args); thr = (ctypes.c_void_p * 2)(g.args[10][0], None); bad[10] = thr
    assert lib.mr_epnp_ransac_grouped(*bad, None) == -1
    bad = list(g.args); bad[21] = 1024
    assert lib.mr_epnp_ransac_grouped(*bad, None) == -1                   # workspace too small for the whole group
    torch.cuda.synchronize()


def test_refit_inside_the_lm_launch_equals_the_two_entry_points(dev):
    """MR_EPNP_DEFER_REFIT + mr_pnp_uncert_from_epnp_grouped (what pnp_uncert / PnPUncert / u2d_pnp_cpu run by default): the initialiser stops
    before its last launch and the LM launch computes the re-fit's pose candidates on the tile it loads anyway — every output, the
    initialiser's hand-over included, equals mr_epnp_ransac_batched followed by mr_pnp_uncert_from_init_batched bit for bit: both layouts,
    fp32 / fp16 / fp64 storage, 2 / 4 / 8 waves per object (the re-fit's three candidates on 2 waves: one wave takes two), per-object
    cameras, plain EPnP (no thresholds), failures, five-candidate objects, NaN / degenerate inputs, P = 4; and the argument checks."""
    import ctypes
    from monorun_amd import _lib
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device, pnp_uncert_from_init_device, pnp_uncert_epnp_device
    lib = _lib.load()
    rng = np.random.default_rng(11)
    def both(x, flags=0, **kw):
        ini, im, iv, _, _ = epnp_ransac_device(x[0], x[1], x[2], x[3], epnp_istd_thres=0.6, epnp_ransac_thres=x[6])
        two = pnp_uncert_from_init_device(*x[:6], ini, im, iv, z_min=0.5, inlier_opt_only=True, flags=flags, with_diag=True)
        one = pnp_uncert_epnp_device(*x[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=x[6], inlier_opt_only=True, flags=flags, with_diag=True)
        torch.cuda.synchronize()
        for k in range(6):
            assert torch.equal(one[k], two[k]) or (one[k].dtype.is_floating_point and torch.equal(torch.nan_to_num(one[k], nan=-7.0), torch.nan_to_num(two[k], nan=-7.0))), (k, kw)
        assert torch.equal(one[6], ini) and torch.equal(one[7], iv), kw
        return one
    for planar in (True, False):
        for dtype in (np.float32, np.float16, np.float64):
            b = syn.make_batch(B=96, hw=(12 if dtype == np.float16 else 28), seed=int(rng.integers(1 << 30)))
            x = [np.array(a, copy=True) for a in syn.pnp_boundary(b, planar=planar)]
            P = x[0].shape[1]
            for i in range(0, 12):                         # gross outliers: many RANSAC iterations
                bad = rng.random(P) < 0.5
                x[2][i, bad] += rng.normal(0, 0.8, (int(bad.sum()), 3)).astype(np.float32)
            x[6][12:16] = 1e-4                             # RANSAC fails
            for i in range(16, 20):                        # exactly five candidates
                x[1][i] = 1e-3; x[1][i, rng.choice(P, 5, replace=False)] = 1.0
            x[2][20] = 0.0; x[0][21, 3] = np.nan; x[2][22, :, 1] = 0.0
            x[3] = np.repeat(x[3].reshape(1, 3, 3), 96, 0) * np.linspace(0.97, 1.03, 96, dtype=np.float32)[:, None, None]
            d = [_t(dev, np.ascontiguousarray(a.astype(dtype)) if k < 3 else a) for k, a in enumerate(x)]
            for w in (0, 2, 4, 8):
                one = both(d, flags=w << 8, planar=planar, dtype=dtype, w=w)
            assert int(one[0].sum()) > 60 and not bool(one[7][12:16].any())
    x = [_t(dev, a) for a in syn.pnp_boundary(syn.make_batch(B=64, seed=5), planar=True)]
    x[6] = None                                            # plain cv2.solvePnP(EPNP) on the istd candidates
    both(x, mode='plain')
    x4 = [_t(dev, np.ascontiguousarray(a[:, :4]) if a.ndim == 3 and k < 3 else a) for k, a in enumerate(syn.pnp_boundary(syn.make_batch(B=8, seed=6), planar=False))]
    both(x4, mode='P=4')
    # argument checks: deferring needs the caller's workspace; the LM side needs it too, large enough, and 1..4 calls
    x = [_t(dev, a) for a in syn.pnp_boundary(syn.make_batch(B=32, seed=7), planar=True)]
    st = lambda t: (ctypes.c_int64 * 3)(*t.stride())
    ip, im, iv = torch.empty(32, 4, device=dev, dtype=torch.float64), torch.empty(32, x[0].shape[1], device=dev, dtype=torch.uint8), torch.empty(32, device=dev, dtype=torch.uint8)
    head = [x[0].data_ptr(), st(x[0]), x[1].data_ptr(), st(x[1]), x[2].data_ptr(), st(x[2]), 0, x[3].data_ptr(), 1]
    assert lib.mr_epnp_ransac_batched(*head, x[6].data_ptr(), 32, x[0].shape[1], 0.6, _lib.MR_EPNP_DEFER_REFIT, 30, ip.data_ptr(), im.data_ptr(), iv.data_ptr(), None, None, None, 0, None) == -1
    one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr() if t is not None else None)
    work = torch.empty(int(lib.mr_epnp_workspace_bytes(32, x[0].shape[1])), device=dev, dtype=torch.uint8)
    outs = [torch.empty(32, device=dev, dtype=torch.uint8), torch.empty(32, 4, device=dev), torch.empty(32, 16, device=dev), torch.empty(32, device=dev), torch.empty(32, x[0].shape[1], device=dev, dtype=torch.uint8)]
    def lm(ncalls=1, wptr=work.data_ptr(), wbytes=work.numel()):
        return lib.mr_pnp_uncert_from_epnp_grouped(ncalls, one(x[0]), st(x[0]), one(x[1]), st(x[1]), one(x[2]), st(x[2]), 0, one(x[3]), 1, one(x[4]), one(x[5]), 1,
                                                   one(ip), one(im), one(iv), None, 32, x[0].shape[1], 0.5, 1, 0, *[one(o) for o in outs], one(None), None, 0.0, None, wptr, wbytes, None)
    assert lm(ncalls=0) == -1 and lm(ncalls=9) == -1 and lm(wptr=None) == -1 and lm(wbytes=work.numel() - 1) == -1
    torch.cuda.synchronize()


def test_fp64_storage_gives_the_fp32_results(dev):
    """fp64 correspondence tensors holding float32 values: the initialiser reads correspondences as float32 (as the reference hands
    them to cv2), so every output equals the fp32-storage run's — both layouts, through the initialiser and the LM launch."""
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device, pnp_uncert_from_init_device
    b = syn.make_batch(B=48, hw=12, seed=31)
    for planar in (True, False):
        x2d, istd, x3d, K, ur, vr, thr = [_t(dev, a) for a in syn.pnp_boundary(b, planar=planar)]
        up = lambda t: torch.empty_strided(t.shape, t.stride(), dtype=torch.float64, device=dev).copy_(t)
        a32 = epnp_ransac_device(x2d, istd, x3d, K, epnp_istd_thres=0.6, epnp_ransac_thres=thr, with_diag=True)
        a64 = epnp_ransac_device(up(x2d), up(istd), up(x3d), K, epnp_istd_thres=0.6, epnp_ransac_thres=thr, with_diag=True)
        torch.cuda.synchronize()
        assert all(torch.equal(p, q) for p, q in zip(a32[:4], a64[:4])), planar
        assert int(a32[2].sum()) >= 40
        o32 = pnp_uncert_from_init_device(x2d, istd, x3d, K, ur, vr, *a32[:3], z_min=0.5, inlier_opt_only=True)
        o64 = pnp_uncert_from_init_device(up(x2d), up(istd), up(x3d), K, ur, vr, *a64[:3], z_min=0.5, inlier_opt_only=True)
        torch.cuda.synchronize()
        assert torch.equal(o32[0], o64[0]) and torch.equal(o32[4], o64[4]) and torch.allclose(o32[1], o64[1], rtol=0, atol=1e-6), planar


def test_reference_flow_on_more_than_65535_objects_in_one_call(dev):
    """One call over 73 728 objects (no grid dimension of the initialiser's launches is limited to 65 535 workgroups; a 5.3 GB workspace):
    initial poses, RANSAC masks, final poses, covariances and masks of every object equal those of 1 024-object calls on the same data, bit
    for bit (both with four waves per object in the LM launch: left to itself the library picks the waves per object by batch size, and
    the lanes' summation order — hence the last bit of a pose now and then — follows it)."""
    from monorun_amd import PnPEpnpLaunch, _lib
    parts = [[torch.from_numpy(np.asarray(a)).to(dev) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=900 + i), planar=False)] for i in range(4)]
    w4 = 4 << _lib.MR_WAVES_SHIFT
    small = [PnPEpnpLaunch(*b[:6], epnp_ransac_thres=b[6], inlier_opt_only=True, flags=w4) for b in parts]
    for l in small:
        l.run()
    reps = 18                                                      # 4 x 18 x 1024 = 73 728 objects
    cat = lambda j: torch.cat([parts[i % 4][j] for i in range(4 * reps)], 0)
    big = PnPEpnpLaunch(cat(0), cat(1), cat(2), parts[0][3], parts[0][4], parts[0][5], epnp_ransac_thres=cat(6), inlier_opt_only=True, flags=w4)
    big.run()
    torch.cuda.synchronize()
    assert big.pose.shape[0] == 73728 and int(big.valid.sum()) > 70000
    for i in range(4 * reps):
        s, l = slice(1024 * i, 1024 * (i + 1)), small[i % 4]
        for n in ('init_pose', 'init_mask', 'init_valid', 'pose', 'mask', 'valid'):
            assert torch.equal(getattr(big, n)[s], getattr(l, n)), (i, n)
        assert torch.allclose(big.cov[s], l.cov, rtol=1e-5, atol=0.0), i


def test_refit_normalisation_switch_mirrors_the_restatement(dev, orc):
    """oracle/epnp.inc, version-dependent decision (i): solvePnPRansac's final re-fit sees float64 normalised image points (default
    since round 4) or float32 ones (round 3's reading, MR_EPNP_REFIT_F32 / orc.set_epnp_refit_f64(False)).  The kernel follows the
    restatement either way: RANSAC masks identical (they are decided before the re-fit), start pose to INIT_TOL; and the two readings
    differ from each other (the switch is live) by far less than the 1e-4 bar after the LM."""
    from monorun_amd import _lib
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device
    b = syn.make_batch(B=48, seed=913)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    d = [_t(dev, a) for a in (x2d, istd, x3d, K, thr)]
    got = {}
    for f64 in (True, False):
        orc.set_epnp_refit_f64(f64)
        try:
            refs = _stage_reference(orc, x2d, istd, x3d, K, thr)
        finally:
            orc.set_epnp_refit_f64(True)
        gpu = epnp_ransac_device(d[0], d[1], d[2], d[3], epnp_istd_thres=0.6, epnp_ransac_thres=d[4], with_diag=True, debug_hypotheses=True,
                                 flags=0 if f64 else _lib.MR_EPNP_REFIT_F32)
        torch.cuda.synchronize()
        _check_stage(gpu, refs)
        got[f64] = gpu[0].cpu().numpy()
    dd = np.abs(got[True] - got[False]).max()
    assert 0.0 < dd < 1e-2, dd


def test_library_wave_rule_is_exported(dev):
    """mr_pick_waves: the library's own rule for waves per object, which PnPPipeline.flags_for applies to all objects in flight."""
    from monorun_amd import _lib, PnPPipeline
    lib = _lib.load()
    with torch.cuda.device(dev):
        assert lib.mr_pick_waves(1024, 784) == 4 and lib.mr_pick_waves(4096, 784) == 2 and lib.mr_pick_waves(0, 784) < 0
    pipe = PnPPipeline(dev, depth=4, verify=False)
    assert pipe.flags_for(1024, 784) == 2 << _lib.MR_WAVES_SHIFT


@pytest.mark.parametrize('flow,waves', [('epnp', 4), ('epnp', 2), ('epnp', 8), ('k0', 4), ('k0', 1), ('k0', 2)])
def test_valid_flag_is_bit_exact_on_adversarial_objects(dev, orc, flow, waves):
    """on planar / collinear / coincident / one-point objects the covariance Hessian J^T J (pnp_uncert.py:71-85,
    hessian.py:84-86) is singular to rounding, and whether its Cholesky factorisation succeeds — the `valid` flag — used to depend on the order
    of its sums (kernel: lane order; restatement: sequential; 2 of 135 227 objects in profiles/r05_epnp_fuzz_long.txt).  Since round 6 the
    covariance stage is a SPECIFIED computation (csrc/pnp_kernel_body.inc stage 4 = oracle orc_cov_hessian_spec: no contraction, IEEE
    division, the workgroup's own summation tree for its wave count): `valid` is compared bit for bit here, on every object, for the wave
    counts the library launches (tests/sweeps/gpu_epnp_fuzz.py is the 2 000-trial version: profiles/r06_epnp_fuzz_long.txt)."""
    from monorun_amd import _lib
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device, pnp_uncert_epnp_device
    from tests import fuzz_cases
    rng = np.random.default_rng(100 + waves)
    fl = waves << _lib.MR_WAVES_SHIFT
    orc.set_cov_waves(waves)
    n_obj = n_singular = n_other_pose = 0
    try:
        for trial in range(40):
            mode = ('planar', 'collinear', 'coincident', 'one_point', 'garbage', 'planar', 'collinear', 'zero_threshold')[trial % 8]
            x2d, istd, x3d, K, ur, vr, thr = fuzz_cases.make_case(mode, rng, hw=(28 if trial % 3 == 0 else None))
            with np.errstate(all='ignore'):
                ref = (orc.u2d_pnp_epnp if flow == 'epnp' else orc.u2d_pnp)(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, num_threads=0)
            d = [_t(dev, a) for a in (x2d, istd, x3d, K, ur, vr, thr)]
            if flow == 'epnp':
                out = pnp_uncert_epnp_device(d[0], d[1], d[2], d[3], d[4], d[5], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=d[6], inlier_opt_only=True, flags=fl)
            else:
                out = pnp_uncert_device(d[0], d[1], d[2], d[3], d[4], d[5], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=d[6], inlier_opt_only=True, flags=fl)
            torch.cuda.synchronize()
            valid, pose, cov, mask = out[0].cpu().numpy().astype(bool), out[1].cpu().numpy(), out[2].cpu().numpy(), out[4].cpu().numpy().astype(bool)
            assert np.array_equal(mask, ref[5]), (trial, mode, 'inlier mask')
            rp = np.concatenate([ref[1], ref[2]], 1)
            same = ((pose == rp) | (np.isnan(pose) & np.isnan(rp))).all(1)       # identical start + identical LM trajectory: the covariance stage sees the same float32 pose
            n_other_pose += int((~same).sum())
            assert np.array_equal(valid[same], ref[0][same]), (trial, mode, 'valid differs for objects', np.flatnonzero(same & (valid != ref[0]))[:8])
            big = np.abs(cov).reshape(len(valid), -1).max(1) > 1e8
            n_obj += len(valid); n_singular += int((big | ~valid).sum())
    finally:
        orc.set_cov_waves(4)
    assert n_singular > 20, 'the cases are meant to contain objects whose Hessian is singular to rounding'
    assert n_other_pose <= 0.01 * n_obj, (n_other_pose, n_obj)


def test_config5_full_size_through_the_default_flow(dev, orc):
    """BASELINE config 5 at its FULL size — 65 536 proposals x 56x56 correspondences, fp16 storage (2.9 GB of inputs) —
    through the flow the boundary runs by default (the reference's initialiser + LM: `pnp_uncert_epnp_device`), in ONE call on one GPU
    (tests/test_gpu_parity.py::test_config5_full_size_on_one_gpu is the same launch through the K0 fast mode): 256 distinct objects tiled
    256x; the first tile against `orc.u2d_pnp_epnp` object by object (masks bit-exact, identical LM iteration counts, pose within 1e-4),
    every other tile bit-equal to the first (results do not depend on batch position or on what else is in flight)."""
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_epnp_device
    nd, rep, hw = 256, 256, 56
    b = syn.make_batch(B=nd, hw=hw, seed=4321)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    mk = lambda a: np.ascontiguousarray(a.astype(np.float16).transpose(0, 2, 1))                     # (nd, C, P) fp16, channel-planar
    t = lambda a: torch.from_numpy(a).to(dev)
    big = [t(mk(a)).repeat(rep, 1, 1).permute(0, 2, 1) for a in (x2d, istd, x3d)]                    # (65536, P, C) views, strides (C*P, 1, P)
    assert big[0].shape == (nd * rep, hw * hw, 2) and big[0].stride() == (2 * hw * hw, 1, hw * hw)
    out = pnp_uncert_epnp_device(big[0], big[1], big[2], t(np.asarray(K)), t(np.asarray(ur)), t(np.asarray(vr)), z_min=0.5, epnp_istd_thres=0.6,
                                 epnp_ransac_thres=t(np.asarray(thr)).repeat(rep), inlier_opt_only=True, with_diag=True)
    torch.cuda.synchronize()
    valid, pose, cov, tr, mask, diag = out[:6]
    f32 = lambda a: np.ascontiguousarray(mk(a).astype(np.float32).transpose(0, 2, 1))              # the oracle sees the rounded values
    ref = orc.u2d_pnp_epnp(f32(x2d), f32(istd), f32(x3d), K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, num_threads=0)
    r_ret, r_yaw, r_t, r_cov, r_tr, r_mask, r_diag = ref
    v0, p0, c0, m0, d0 = [a[:nd].cpu().numpy() for a in (valid, pose, cov, mask, diag)]
    assert np.array_equal(m0.astype(bool), r_mask) and np.array_equal(v0.astype(bool), r_ret)
    assert np.array_equal(d0[:, 0], r_diag[:, 0]) and np.array_equal(d0[:, 2] % 16, r_diag[:, 2]), 'LM iteration counts / exit reasons'
    ok = r_ret
    dyaw = np.abs(np.angle(np.exp(1j * (p0[:, 0] - r_yaw[:, 0]))))
    assert dyaw[ok].max() <= POSE_TOL and np.abs(p0[:, 1:] - r_t)[ok].max() <= POSE_TOL
    scale = np.abs(r_cov[ok]).reshape(ok.sum(), -1).max(1)[:, None, None]
    assert (np.abs(c0[ok] - r_cov[ok]) / scale).max() <= 1e-5
    for v in (valid, pose, cov, tr, mask):
        tiles = v.view(rep, nd, *v.shape[1:])
        assert bool((tiles == tiles[:1]).all()), 'a later tile differs from the first'
    assert float(valid.float().mean()) > 0.97


def test_a_dropped_launch_frees_its_buffers_by_refcount(dev):
    """PnPEpnpLaunch(fused=True) used to hold the launch set of its one call, whose `members` held the launch — a reference
    cycle, so the ~17 MB-per-1024-object workspace, the outputs and the masks waited for the cyclic garbage collector.  Now the launch set
    keeps argument lists only: dropping the last reference frees the tensors at once."""
    import gc
    import weakref
    from monorun_amd import PnPEpnpLaunch
    b = syn.make_batch(B=64, seed=5)
    d = [_t(dev, a) for a in syn.pnp_boundary(b, planar=True)]
    gc.collect()
    gc.disable()
    try:
        l = PnPEpnpLaunch(*d[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=d[6], inlier_opt_only=True)
        l.run()
        torch.cuda.synchronize()
        refs = [weakref.ref(l), weakref.ref(l.work), weakref.ref(l.mask)]
        del l
        assert all(r() is None for r in refs), 'a reference cycle keeps the launch (and its device buffers) alive until the cyclic GC runs'
    finally:
        gc.enable()


def test_opencv_early_return_switch_mirrors_the_restatement(dev, orc):
    """oracle/epnp.inc decision (ii) behind a switch on both sides: with EXACTLY five istd candidates OpenCV >= 3.3 returns
    solvePnP(EPNP) on the float32 inputs; MR_EPNP_CV_EARLY_RETURN / orc.set_epnp_cv_early_return(True) restate that (float32 normalisation of the
    image points), the default keeps the float64 normalisation of every re-fit.  The kernel follows the restatement either way (start pose to
    INIT_TOL, all five points inliers), the two settings differ from each other (the switch is live), and objects with more candidates do not
    notice it."""
    from monorun_amd import _lib
    from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device
    b = syn.make_batch(B=24, hw=8, seed=31)
    x2d, istd, x3d, K, ur, vr, thr = [np.array(a, copy=True) for a in syn.pnp_boundary(b, planar=False)]
    rng = np.random.default_rng(2)
    five = np.arange(24) % 2 == 0
    for i in np.flatnonzero(five):                        # exactly five points pass the istd test (0.6 x the mean of each axis)
        keep = rng.choice(x2d.shape[1], 5, replace=False)
        istd[i] = 0.001
        istd[i, keep] = 1.0
    d = [_t(dev, a) for a in (x2d, istd, x3d, K, thr)]
    got = {}
    for early in (False, True):
        orc.set_epnp_cv_early_return(early)
        try:
            refs = _stage_reference(orc, x2d, istd, x3d, K, thr)
        finally:
            orc.set_epnp_cv_early_return(False)
        assert all(r['n'] == 5 for r, f in zip(refs, five) if f)
        gpu = epnp_ransac_device(d[0], d[1], d[2], d[3], epnp_istd_thres=0.6, epnp_ransac_thres=d[4], with_diag=True, debug_hypotheses=True,
                                 flags=_lib.MR_EPNP_CV_EARLY_RETURN if early else 0)
        torch.cuda.synchronize()
        _check_stage(gpu, refs)
        got[early] = gpu[0].cpu().numpy()
    dd = np.abs(got[True] - got[False])
    # (five noisy points determine a pose badly: the two normalisations can land far apart — what matters is that the kernel lands where the restatement does)
    assert dd[five].max() > 0.0 and dd[~five].max() == 0.0, (dd[five].max(), dd[~five].max())


def test_wide_launches_equal_the_quad_launches(dev):
    """Round 6: launches that would leave most SIMDs without a wave map a hypothesis / a re-fit to a 16-lane row or to a whole wave instead of a quad
    (csrc/epnp_stages.inc epnp_hyp_kernel<LV>, epnp_refit_betas_kernel<LV>): the extra quads take LV levels of the eigen-solver's bisection per round.
    Same midpoints, same comparisons: EVERY output of the flow — valid, pose, covariance, radius, masks, start poses, diagnostics — is bit-identical
    for the three mappings (the library reads MR_EP_WIDE once per process, so each mapping runs in its own; sha256 over all outputs of all batches)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for lv in ('0', '2', '4'):
        env = dict(os.environ, MR_EP_WIDE=lv, OBJECTS='37,300', NBATCH='3')
        out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'gpu_wide_ab.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600).stdout.decode()
        rows = [l for l in out.splitlines() if 'sha256 of all outputs' in l]
        assert len(rows) == 2, out[-2000:]
        digests[lv] = [r.split('sha256 of all outputs')[1].strip() for r in rows]
    assert digests['0'] == digests['2'] == digests['4'], digests

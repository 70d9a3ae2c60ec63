"""Pin the oracle's LM (R3) independently of Ceres: known-answer cube (config 1), scipy minimiser,
and the Ceres-1.14 control-flow properties that the restatement must exhibit.  CPU only."""
import numpy as np
import pytest
from scipy.optimize import least_squares

from monorun_amd import synthetic as syn


def _cost_fn(orc, K, clips, p2, p3, w):
    def f(x):
        r, _ = orc.residual_jacobian(K, clips, x, p2, p3, w)
        return r.ravel()
    def j(x):
        _, J = orc.residual_jacobian(K, clips, x, p2, p3, w)
        return J.reshape(-1, 4)
    return f, j


def test_config1_noise_free_cube_recovers_gt(orc):
    c = syn.cube_config1()
    r = orc.pnp_uncert(c['pts2d'], c['pts3d'], c['wgt2d'], c['K'], c['init_pose'], c['clips'], with_cov=True)
    assert r['val'] == 1
    assert np.abs(r['pose'] - c['gt_pose']).max() < 1e-6          # stops on Ceres' parameter tolerance, not at 0
    assert r['final_cost'] < 1e-9 * r['initial_cost']
    assert r['why'] in (1, 2, 3) and 2 <= r['iters'] <= 10
    assert np.all(np.linalg.eigvalsh(r['cov']) > 0) and np.allclose(r['cov'], r['cov'].T)
    # starting AT the solution: gradient tolerance fires at iteration 0, radius stays at its initial 1e4
    r0 = orc.pnp_uncert(c['pts2d'], c['pts3d'], c['wgt2d'], c['K'], c['gt_pose'], c['clips'])
    assert r0['val'] == 1 and r0['iters'] == 0 and r0['why'] == 1 and r0['tr'] == 1e4
    assert np.array_equal(r0['pose'], c['gt_pose'])


def test_lm_agrees_with_scipy_minimiser(orc, batch64):
    """The Ceres-style stopping iterate must sit within function-tolerance distance of the true
    minimiser of the same cost (found by scipy's LM at machine tolerances)."""
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=False)
    K = K[0].astype(np.float64)
    clips = np.array([0.5, ur[0, 0], ur[0, 1], vr[0, 0], vr[0, 1]], np.float64)
    worst_cost_gap, worst_pose = 0.0, 0.0
    for b in range(12):
        ok = ~batch64['outlier'][b].ravel()
        p2, p3, w = x2d[b][ok].astype(np.float64), x3d[b][ok].astype(np.float64), istd[b][ok].astype(np.float64)
        gt = np.array([batch64['gt_yaw'][b], *batch64['gt_t'][b]])
        init = gt + np.array([0.1, 0.3, 0.1, 1.0])
        r = orc.pnp_uncert(p2, p3, w, K, init, clips)
        f, j = _cost_fn(orc, K, clips, p2, p3, w)
        s = least_squares(f, r['pose'], jac=j, method='lm', xtol=1e-15, ftol=1e-15, gtol=1e-15)
        assert r['val'] == 1
        cmin = 0.5 * np.sum(s.fun ** 2)
        assert r['final_cost'] >= cmin * (1 - 1e-12)
        worst_cost_gap = max(worst_cost_gap, (r['final_cost'] - cmin) / cmin)
        worst_pose = max(worst_pose, np.abs(r['pose'] - s.x).max())
    assert worst_cost_gap < 1e-5          # function_tolerance = 1e-6 on the last *discarded* step
    assert worst_pose < 5e-2


def test_lm_control_flow_properties(orc, batch64):
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=False)
    K = K[0].astype(np.float64)
    clips = np.array([0.5, ur[0, 0], ur[0, 1], vr[0, 0], vr[0, 1]], np.float64)
    b = 3
    gt = np.array([batch64['gt_yaw'][b], *batch64['gt_t'][b]])
    r = orc.pnp_uncert(x2d[b], x3d[b], istd[b], K, gt + np.array([0.3, 1.0, 0.3, 3.0]), clips)
    assert r['val'] == 1 and r['final_cost'] <= r['initial_cost']
    assert 1 <= r['iters'] <= 50 and r['n_success'] <= r['iters']
    assert r['tr'] > 0
    # NaN input -> evaluation failure at iteration 0 -> unusable, pose = init
    bad = x3d[b].astype(np.float64).copy(); bad[5, 1] = np.nan
    rb = orc.pnp_uncert(x2d[b], bad, istd[b], K, gt, clips)
    assert rb['val'] == 0 and rb['why'] == 7 and np.array_equal(rb['pose'], gt)
    # zero weights: zero gradient -> converged immediately at the initial pose
    rz = orc.pnp_uncert(x2d[b], x3d[b], 0 * istd[b], K, gt, clips, with_cov=True)
    assert rz['iters'] == 0 and rz['why'] == 1 and np.array_equal(rz['pose'], gt)
    assert rz['val'] == 0 and np.array_equal(rz['cov'], np.eye(4))       # rank-deficient covariance -> val 0, cov untouched


def test_max_iterations_gives_usable_no_convergence(orc):
    """A hard problem (far init, everything clamped at the border half of the time) may run out of
    iterations; Ceres reports NO_CONVERGENCE which IsSolutionUsable() accepts."""
    c = syn.cube_config1(n_points=64, seed=3)
    init = c['gt_pose'] + np.array([2.5, 9.0, -3.0, 25.0])
    r = orc.pnp_uncert(c['pts2d'], c['pts3d'], c['wgt2d'], c['K'], init, c['clips'])
    assert r['termination'] in (0, 1) and r['val'] == 1 and r['iters'] <= 50
    if r['termination'] == 1:
        assert r['iters'] == 50 and r['why'] == 4


def test_batch_driver_matches_per_object_calls(orc, batch64):
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=False)
    out = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True)
    ret, yaw, t, cov, tr, mask, diag = out
    assert ret.dtype == bool and yaw.shape == (64, 1) and t.shape == (64, 3) and cov.shape == (64, 4, 4)
    assert tr.shape == (64, 1) and mask.shape == (64, 784) and mask.dtype == bool
    assert ret.all()
    # multi-threaded == single-threaded (objects are independent)
    out_mt = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, num_threads=0)
    for a, b in zip(out[:6], out_mt):
        assert np.array_equal(a, b)
    # object 0 by hand: mask -> k0 -> lm on inliers -> float32 -> torch hessian -> inverse
    m0 = orc.istd_inlier_mask(x2d[:1] * 0 + istd[:1], 0.6)[0]
    k0 = orc.k0_init(x2d[0], x3d[0], m0, K[0], thr[0])
    assert k0['ok'] and np.array_equal(k0['mask'], mask[0])
    sel = k0['mask']
    clips = np.array([0.5, ur[0, 0], ur[0, 1], vr[0, 0], vr[0, 1]], np.float64)
    r = orc.pnp_uncert(x2d[0][sel], x3d[0][sel], istd[0][sel], K[0], k0['init_pose'], clips)
    assert np.array_equal(r['pose'].astype(np.float32), np.concatenate([yaw[0], t[0]]))
    assert np.float32(r['tr']) == tr[0, 0] and r['iters'] == diag[0, 0]
    _, _, H = orc.torch_jacobian(K[0], 0.5, ur[0], vr[0], yaw[0, 0], t[0], x2d[0], x3d[0], istd[0], sel)
    ok, c = orc.pose_cov(H)
    assert ok and np.array_equal(c.astype(np.float32), cov[0])


def test_empty_batch_and_broadcast_forms(orc, batch64):
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=False)
    e = orc.u2d_pnp(x2d[:0], istd[:0], x3d[:0], K, ur, vr)
    assert [a.shape for a in e] == [(0,), (0, 1), (0, 3), (0, 4, 4), (0, 1), (0, 784)]
    a = orc.u2d_pnp(x2d[:4], istd[:4], x3d[:4], K, ur, vr, 0.5, 0.6, thr[:4], True)
    b = orc.u2d_pnp(x2d[:4], istd[:4], x3d[:4], np.repeat(K, 4, 0), np.repeat(ur, 4, 0), np.repeat(vr, 4, 0), 0.5, 0.6, thr[:4], True)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)

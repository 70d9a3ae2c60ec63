"""True 6-DoF mode (`use_6dof`, which the reference declares and ignores, pnp_uncert.py:11).
No reference code exists for it, so the CPU restatement (oracle: the reference's residual functor on Jets with 6 partials +
the same Ceres LM) is pinned independently — finite differences, scipy.optimize.least_squares, a known 6-DoF answer —
and the HIP kernel (closed-form Jacobian through the left Jacobian of SO(3)) is then compared with it."""
import numpy as np
import pytest
from scipy.optimize import least_squares

from monorun_amd import synthetic as syn

K9 = np.array([707.0912, 0, 601.8873, 0, 707.0912, 183.1104, 0, 0, 1.0])
CLIPS = np.array([0.5, -200, 1442, -200, 575.])


def _rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * kx + (1 - np.cos(th)) * kx @ kx


def _scene(seed, n=120, noise=0.5):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, 3)) * np.array([2, 0.8, 0.9])
    pose = np.array([0.08, -0.7, 0.05, 1.0, 1.5, 14.0])
    xc = X @ _rodrigues(pose[:3]).T + pose[3:]
    uv = np.stack([K9[0] * xc[:, 0] / xc[:, 2] + K9[2], K9[4] * xc[:, 1] / xc[:, 2] + K9[5]], 1) + rng.normal(0, noise, (n, 2))
    w = rng.uniform(0.05, 0.3, (n, 2))
    return X, uv, w, pose


def test_jet_jacobian_against_finite_differences_and_small_angle_branch(orc):
    X, uv, w, pose = _scene(1)
    for p in (pose, np.array([1e-10, 0, -1e-10, 1.0, 1.5, 14.0]), np.array([0.0, 0.0, 0.0, 0.5, 1.0, 9.0])):     # Rodrigues and first-order branches
        ok, c, g, H, res, jac = orc.eval6(uv, X, w, K9, p, CLIPS)
        assert ok and np.allclose(g, np.einsum('nrk,nr->k', jac, res)) and np.allclose(H, np.einsum('nrk,nrl->kl', jac, jac))
        for k in range(6):
            e = np.zeros(6); e[k] = 1e-6
            rp = orc.eval6(uv, X, w, K9, p + e, CLIPS)[4]; rm = orc.eval6(uv, X, w, K9, p - e, CLIPS)[4]
            assert np.abs((rp - rm) / 2e-6 - jac[:, :, k]).max() <= 2e-5 * max(1.0, np.abs(jac[:, :, k]).max())
    # a pure yaw pose: columns ry, tx, ty, tz of the 6-DoF Jacobian are the 4-DoF Jacobian (R1)
    p4 = np.array([0.4, 1.0, 1.5, 14.0])
    jac6 = orc.eval6(uv, X, w, K9, np.array([0, p4[0], 0, *p4[1:]]), CLIPS)[5]
    _, jac4 = orc.residual_jacobian(K9, CLIPS, p4, uv, X, w)
    assert np.abs(jac6[:, :, [1, 3, 4, 5]] - jac4).max() <= 1e-9


def test_lm_solution_against_scipy_and_known_answer(orc):
    X, uv, w, pose = _scene(2, noise=0.0)
    r = orc.pnp6_uncert(uv, X, w, K9, pose + np.array([0.05, 0.1, -0.05, 0.3, 0.1, 1.0]), CLIPS)
    assert r['val'] == 1 and np.abs(r['pose'] - pose).max() <= 1e-6                                     # noise-free: the generating pose
    X, uv, w, pose = _scene(3, noise=1.0)
    r = orc.pnp6_uncert(uv, X, w, K9, pose + np.array([0.05, 0.1, -0.05, 0.3, 0.1, 1.0]), CLIPS)
    f = lambda p: orc.eval6(uv, X, w, K9, p, CLIPS)[4].ravel()
    sp = least_squares(f, r['pose'], method='lm', xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert r['val'] == 1 and np.abs(sp.x - r['pose']).max() <= 5e-4 and abs(0.5 * (sp.fun ** 2).sum() - r['final_cost']) <= 1e-6 * r['final_cost']
    H = orc.eval6(uv, X, w, K9, r['pose'], CLIPS)[3]
    assert np.allclose(r['cov'] @ H, np.eye(6), atol=1e-8)


@pytest.mark.gpu
def test_hip_6dof_refinement_matches_the_oracle(orc):
    import torch
    from monorun_amd.ops import pnp_uncert, PnPUncert
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device, pnp6_refine_device
    dev = torch.device('cuda:0')
    b = syn.make_batch(B=96, seed=17)
    for planar in (True, False):
        x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=planar)

        def dv(a):
            t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
        args = [dv(a) for a in (x2d, istd, x3d, K, ur, vr)]
        valid4, pose4, cov4, tr, mask, _ = pnp_uncert_device(*args, 0.5, 0.6, dv(thr), True)
        valid6, pose6, cov6, diag = pnp6_refine_device(*args, mask, pose4, valid4, 0.5, with_diag=True)
        torch.cuda.synchronize()
        c = [np.ascontiguousarray(a) for a in (x2d, istd, x3d)]
        rv, rp, rc, rd = orc.pnp6_refine(c[0], c[1], c[2], K, ur, vr, mask.cpu().numpy(), pose4.cpu().numpy(), valid4.cpu().numpy(), num_threads=0)
        assert np.array_equal(valid6.cpu().numpy().astype(bool), rv) and rv.sum() >= 90
        assert np.array_equal(diag.cpu().numpy(), rd), 'LM iteration counts / exit reasons'
        assert np.abs(pose6.cpu().numpy() - rp)[rv].max() <= 1e-4
        sc = np.abs(rc[rv]).reshape(rv.sum(), -1).max(1)[:, None, None]
        assert (np.abs(cov6.cpu().numpy()[rv] - rc[rv]) / sc).max() <= 1e-4
        # the 6-DoF optimum cannot cost more than the 4-DoF one it started from, and stays close to a yaw-only rotation here
        assert np.abs(pose6.cpu().numpy()[:, [0, 2]])[rv].max() <= 0.5
    # through the drop-in API: use_6dof=True returns the angle-axis vector and a 6x6 covariance; False is untouched
    x2d, istd, x3d, K, ur, vr, thr = [dv(a) for a in syn.pnp_boundary(b, planar=True)]
    ret, r_vec, t_vec, cov, m = pnp_uncert(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, use_6dof=True)
    assert r_vec.shape == (96, 3) and t_vec.shape == (96, 3) and cov.shape == (96, 6, 6) and m.shape == (96, 784) and ret.dtype == torch.bool
    ret4, r4, t4, cov4, m4 = PnPUncert(inlier_opt_only=True)(x2d, istd, x3d, K, ur, vr, thr)
    assert r4.shape == (96, 1) and cov4.shape == (96, 4, 4) and torch.equal(m, m4)
    ok = (ret & ret4).cpu().numpy()
    assert np.abs((r_vec[:, 1:2] - r4).cpu().numpy())[ok].max() <= 0.3 and np.abs((t_vec - t4).cpu().numpy())[ok].max() <= 2.0

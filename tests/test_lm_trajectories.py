"""G7 (tests/golden/g7_lm_trajectories.npz): committed LM trajectories of 64 objects — ordinary convergence by every tolerance,
rejected steps, max-iteration exits, an evaluation failure, z/u/v clamps, degenerate inputs.
  * CPU: the oracle reproduces the committed data in BOTH step-solver modes (Cholesky of the normal equations = what the
    HIP kernel does; Householder QR of [J S; D] = what Ceres' DENSE_QR does, pnp_uncert_cpu.cpp:270-274), and the two modes
    agree on a config-2 batch to 1e-10 in the fp64 pose with identical iteration counts / exit reasons.
  * GPU: the kernel, run with max_num_iterations = k for every k, matches the committed trajectory pass by pass (cost and
    trust-region radius after pass k, iteration count, exit reason) and the committed final pose."""
import os

import numpy as np
import pytest

from monorun_amd import synthetic as syn

G7 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g7_lm_trajectories.npz')
IT, COST, CAND, MCC, RD, RAD, SN, OUT = range(8)


@pytest.fixture(scope='module')
def g7():
    return dict(np.load(G7, allow_pickle=True))


def _solve(orc, z, i, **kw):
    clips = np.array([0.5, z['ur'][i, 0], z['ur'][i, 1], z['vr'][i, 0], z['vr'][i, 1]], np.float64)
    return orc.pnp_uncert_opt(z['x2d'][i].astype(np.float64), z['x3d'][i].astype(np.float64), z['w'][i].astype(np.float64),
                              z['K'][i].astype(np.float64), z['init'][i], clips, **kw)


def test_fixture_covers_the_control_flow(g7):
    z = g7
    assert z['x2d'].shape == (64, 196, 2) and set(np.unique(z['why'])) >= {1, 2, 3, 4, 7}
    out = z['trace'][:, :, OUT]
    assert (out == 0).any(1).sum() >= 10, 'rejected steps'
    assert (z['iters'] == 50).sum() >= 3 and (z['iters'] == 0).sum() >= 2
    # 73 of the 1924 candidates were left out: there the normal equations and QR part ways (iteration count, exit reason, or a final
    # pose that differs by more than 1e-7) because J S is numerically rank-deficient — 71 with the initial pose at / behind the camera
    # (every point z-clamped: directions without information), 2 started more than 10 m off
    assert int(z['pool_size']) == 1924 and z['pool_divergent_qr_vs_cholesky'].tolist() == [['far', '2'], ['zclamp', '71']]


@pytest.mark.parametrize('qr', [False, True])
def test_oracle_reproduces_the_committed_trajectories(orc, g7, qr):
    z = g7
    for i in range(64):
        r = _solve(orc, z, i, qr=qr, trace=True)
        assert (r['iters'], r['why'], r['val']) == (z['iters'][i], z['why'][i], z['val'][i]), (i, z['tag'][i])
        n = int(z['n_pass'][i])
        assert len(r['trace']) == n
        a, b = r['trace'], z['trace'][i, :n]
        assert np.array_equal(a[:, OUT], b[:, OUT]) and np.array_equal(np.isnan(a), np.isnan(b))
        # the committed data came from the QR mode; the Cholesky mode agrees to the conditioning of the normal equations
        for col, tol in ((COST, 1e-12), (CAND, 1e-9), (RAD, 1e-9 if qr else 1e-5), (RD, 1e-9 if qr else 1e-4)):
            ok = ~np.isnan(b[:, col])
            atol = 1e-12 * np.nanmax(b[:, COST]) if (col in (COST, CAND) and n) else 1e-300     # costs that reached rounding level are noise
            assert np.allclose(a[ok, col], b[ok, col], rtol=tol if qr else max(tol, 1e-6), atol=atol), (i, z['tag'][i], col)
        if z['val'][i]:
            assert np.abs(r['pose'] - z['pose'][i]).max() <= (1e-9 if qr else 1e-6) * max(1.0, np.abs(z['pose'][i]).max()), (i, z['tag'][i])


def test_qr_and_cholesky_step_solvers_agree_on_config2(orc):
    """the normal equations are a faithful stand-in for DENSE_QR on the workload: identical iteration counts, exit reasons,
    masks; fp64 poses within 1e-10 (204 800-object version: tests/sweeps/lm_qr_vs_chol_sweep.py, result in DESIGN.md §4)."""
    for seed in (1234, 5):
        b = syn.make_batch(B=256, seed=seed)
        x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
        try:
            orc.set_lm_options(qr=False)
            r0 = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, return_pose64=True, num_threads=0)
            orc.set_lm_options(qr=True)
            r1 = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, return_pose64=True, num_threads=0)
        finally:
            orc.set_lm_options()
        assert np.array_equal(r0[6][:, 0], r1[6][:, 0]) and np.array_equal(r0[6][:, 2], r1[6][:, 2]) and np.array_equal(r0[5], r1[5])
        assert np.abs(r0[7] - r1[7]).max() <= 1e-10 and np.array_equal(r0[4], r1[4])


def _replay_on_gpu(z, pose_tol=1e-4, strict=True):
    """run the kernel with max_num_iterations = k for every k and compare with the committed trajectory; returns the number of
    objects that leave the committed path (0 when strict: every object is asserted)"""
    import torch
    from monorun_amd import _lib
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
    dev = torch.device('cuda:0')
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
    x2d, w, x3d, K, ur, vr = t(z['x2d']), t(z['w']), t(z['x3d']), t(z['K']), t(z['ur']), t(z['vr'])
    init = t(z['init'], torch.float64)
    tr_ = z['trace']
    flagged = []

    def run(max_iter):
        flags = _lib.MR_NO_ISTD_MASK | _lib.MR_COV_NONE | (max_iter << _lib.MR_LM_MAXIT_SHIFT)     # no covariance: `valid` is the LM's own verdict
        valid, pose, cov, tr, mask, diag = pnp_uncert_device(x2d, w, x3d, K, ur, vr, z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=None,
                                                             inlier_opt_only=False, init_pose=init, flags=flags, with_diag=True)
        torch.cuda.synchronize()
        d = diag.cpu().numpy()
        flagged.append(d[:, 2].astype(int) // 16)                 # MR_DIAG_WHY_ILL_CONDITIONED rides on the exit reason
        d[:, 2] = d[:, 2] % 16
        return valid.cpu().numpy().astype(bool), pose.cpu().numpy(), tr.cpu().numpy(), d
    # the full run: iteration counts, exit reasons, validity, final pose, final radius
    full = run(0)
    valid, pose, tr, diag = full
    fail = z['why'] == 7
    on_path = (diag[:, 0].astype(int) == z['iters']) & (diag[:, 2].astype(int) == z['why'])
    if strict:
        assert on_path.all()
        assert np.array_equal(valid[~fail], z['val'][~fail].astype(bool))
    ok = z['val'].astype(bool) & on_path
    scale = np.maximum(1.0, np.abs(z['pose']).max(1))
    pose_gap = np.abs(pose - z['pose'].astype(np.float32)).max(1) / scale
    if pose_tol is not None:
        assert (pose_gap[ok] <= pose_tol).all()
    assert np.allclose(tr[ok], z['radius'][ok].astype(np.float32), rtol=1e-5 if strict else 1e-3)
    # truncated runs: after k passes the kernel holds the committed cost and radius of pass k
    for k in range(1, int(z['iters'].max()) + 1):
        valid, pose, tr, diag = run(k) if k < 50 else full
        live = (z['iters'] >= k) & on_path                        # objects that execute a k-th pass
        if not live.any():
            continue
        idx = np.where(live)[0]
        row = tr_[idx, k - 1]
        stopped_here = (z['iters'][idx] == k) & np.isin(row[:, OUT], (2, 3))          # tolerance exit inside pass k: candidate discarded
        exp_cost = np.where(row[:, OUT] == 1, row[:, CAND], row[:, COST])
        assert np.array_equal(diag[idx, 0].astype(int), np.full(len(idx), k)), k
        exp_why = np.where(stopped_here, z['why'][idx], 4)
        exp_why = np.where((z['iters'][idx] == k) & ~stopped_here, z['why'][idx], exp_why)      # e.g. the 50th pass of a max-iteration object
        assert np.array_equal(diag[idx, 2].astype(int), exp_why), k
        with np.errstate(over='ignore'):
            assert np.allclose(diag[idx, 1], exp_cost.astype(np.float32), rtol=2e-6 if strict else 1e-4), k
        assert np.allclose(tr[idx], row[:, RAD].astype(np.float32), rtol=1e-5 if strict else 1e-3), k
    return int((~on_path).sum()), pose_gap, on_path, flagged[0].astype(bool)


@pytest.mark.gpu
def test_kernel_follows_the_committed_trajectories_pass_by_pass(g7):
    off, _, _, flagged = _replay_on_gpu(g7)
    assert off == 0
    print('G7: objects flagged MR_DIAG_WHY_ILL_CONDITIONED:', int(flagged.sum()), 'of', len(flagged))


G7B = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g7b_lm_rank_deficient_starts.npz')


@pytest.fixture(scope='module')
def g7b():
    return dict(np.load(G7B, allow_pickle=True))


def test_g7b_is_the_set_g7_leaves_out_and_the_oracle_reproduces_it(orc, g7, g7b):
    """G7b = the 73 candidates of G7's pool where the two step solvers part ways (rank-deficient J S: starts at / behind the camera or
    > 10 m off).  The fixture freezes the oracle's CHOLESKY-mode trajectories — the step solver the kernel has — and records what
    the QR mode (Ceres' DENSE_QR) returns for the same objects: the gap is committed data (profiles/r03_g7b_qr_vs_cholesky.txt)."""
    z = g7b
    assert len(z['iters']) == 73 and int(z['pool_size']) == int(g7['pool_size'])
    tags, cnt = np.unique(z['tag'], return_counts=True)
    assert dict(zip(tags.tolist(), cnt.tolist())) == {'far': 2, 'zclamp': 71}
    differ = (z['iters'] != z['qr_iters']) | (z['why'] != z['qr_why']) | (np.abs(z['pose'] - z['qr_pose']).max(1) > 1e-7 * np.maximum(1.0, np.abs(z['qr_pose']).max(1)))
    assert differ.all()
    for i in range(73):
        r = _solve(orc, z, i, qr=False, trace=True)
        assert (r['iters'], r['why'], r['val']) == (z['iters'][i], z['why'][i], z['val'][i]), i
        n = int(z['n_pass'][i])
        a, b = r['trace'], z['trace'][i, :n]
        assert len(a) == n and np.array_equal(a[:, OUT], b[:, OUT])
        ok = ~np.isnan(b[:, COST])
        assert np.allclose(a[ok, COST], b[ok, COST], rtol=1e-12) and np.abs(r['pose'] - z['pose'][i]).max() <= 1e-9 * max(1.0, np.abs(z['pose'][i]).max())
        rq = _solve(orc, z, i, qr=True)
        assert (rq['iters'], rq['why']) == (z['qr_iters'][i], z['qr_why'][i])


@pytest.mark.gpu
def test_kernel_on_the_rank_deficient_starts_g7b(g7b):
    """The kernel on the inputs where it is pinned to nothing but its own step solver: it must walk the
    committed Cholesky-mode trajectory pass by pass.  These systems are singular to working precision, so the kernel's reciprocal
    pivots (v_rsq_f64 + two Newton steps instead of IEEE sqrt / div) may legitimately leave the committed path on a few objects:
    those are counted and bounded, every other object is compared pass by pass."""
    z = g7b
    off, pose_gap, on_path, flagged = _replay_on_gpu(z, pose_tol=None, strict=False)
    assert off <= 8, off
    # The POSE of these objects has directions (almost) without information — that is why the step solvers part ways — and along them
    # the returned value is decided by rounding: the kernel's reciprocal pivots move it by up to ~1e-2 even where the oracle's two
    # modes agree to 1e-7, while cost and trust-region radius agree pass by pass (asserted above, 1e-4 / 1e-3).  The pose gaps are
    # therefore REPORTED (profiles/r03_g7b_kernel.txt), not bounded by the 1e-4 bar, which holds where the problem determines the pose.
    ok = z['val'].astype(bool) & on_path
    ref_gap = np.abs(z['pose'] - z['qr_pose']).max(1) / np.maximum(1.0, np.abs(z['pose']).max(1))
    assert np.isfinite(pose_gap[ok]).all() and np.median(pose_gap[ok]) <= 1e-4
    # what a caller can tell (MR_DIAG_WHY_ILL_CONDITIONED, added to diag's exit reason: pivot ratio of the scaled damped normal matrix
    # below 1e-10 in some pass): every object of this fixture carries the flag (their smallest ratio is 1.1e-11; a config-2 batch has
    # none below 2.5e-4 — tests/test_gpu_parity.py asserts that no object there is flagged)
    assert flagged.all()
    if os.environ.get('MR_G7B_REPORT'):
        for i in range(len(ok)):
            print(f'{i:3d} {z["tag"][i]:7s} on_path {bool(on_path[i])} flagged {bool(flagged[i])} kernel-vs-cholesky {pose_gap[i]:.3e} cholesky-vs-qr {ref_gap[i]:.3e}')
        print(f'# on the committed path: {int(on_path.sum())} of {len(on_path)}; kernel-vs-cholesky pose gap (relative to max(1,|pose|)) median {np.median(pose_gap[ok]):.2e} '
              f'p90 {np.percentile(pose_gap[ok], 90):.2e} max {pose_gap[ok].max():.2e}; within 1e-4: {int((pose_gap[ok] <= 1e-4).sum())} of {int(ok.sum())}')

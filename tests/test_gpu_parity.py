"""GPU parity tests (run with -m gpu on an MI355X): the fused HIP kernel, called through the C ABI,
against the CPU oracle on identical inputs, against the committed golden vectors, and — at
BASELINE.json's full size — through size-independent properties.

Bars (north_star): integer / index / mask outputs bit-exact; pose within 1e-4 (rotation and
translation); covariance within 1e-5 relative."""
import ctypes

import numpy as np
import pytest
import torch

from monorun_amd import synthetic as syn

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4
COV_RTOL = 1e-5


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _dv(a, dev):
    t = torch.from_numpy(np.asarray(a))
    d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev)
    d.copy_(t)
    return d


def _run(dev, x2d, istd, x3d, K, ur, vr, thr, init=None, flags=0, z_min=0.5, thres=0.6, inlier_opt_only=True):
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
    out = pnp_uncert_device(_dv(x2d, dev), _dv(istd, dev), _dv(x3d, dev), _dv(K, dev), _dv(ur, dev), _dv(vr, dev),
                            z_min=z_min, epnp_istd_thres=thres, epnp_ransac_thres=_dv(thr, dev) if thr is not None else None,
                            inlier_opt_only=inlier_opt_only, init_pose=_dv(init, dev) if init is not None else None,
                            flags=flags, with_diag=True)
    torch.cuda.synchronize()
    valid, pose, cov, tr, mask, diag = [t.cpu().numpy() for t in out]
    return valid.astype(bool), pose, cov, tr, mask.astype(bool), diag


def _cmp(gpu, ref, tag=''):
    valid, pose, cov, tr, mask, diag = gpu
    r_ret, r_yaw, r_t, r_cov, r_tr, r_mask, r_diag = ref
    assert np.array_equal(mask, r_mask), f'{tag}: inlier mask differs in {(mask != r_mask).sum()} points'
    assert np.array_equal(valid, r_ret), tag
    assert np.array_equal(diag[:, 0], r_diag[:, 0]) and np.array_equal(diag[:, 2] % 16, r_diag[:, 2]), f'{tag}: LM iteration count / reason'     # % 16: MR_DIAG_WHY_ILL_CONDITIONED rides on the reason
    assert np.array_equal(diag[:, 3], r_diag[:, 3]), f'{tag}: K0 consensus size'
    dyaw = np.abs(np.angle(np.exp(1j * (pose[:, 0] - r_yaw[:, 0]))))
    assert dyaw.max() <= POSE_TOL and np.abs(pose[:, 1:] - r_t).max() <= POSE_TOL, (tag, dyaw.max(), np.abs(pose[:, 1:] - r_t).max())
    # covariance: compared where the solve is valid (an invalid object's J^T J at the zero pose can be
    # numerically singular, where (J^T J)^-1 is summation-order noise on both sides)
    ok = r_ret
    if ok.any():
        scale = np.abs(r_cov[ok]).reshape(ok.sum(), -1).max(1)[:, None, None]
        assert (np.abs(cov[ok] - r_cov[ok]) / scale).max() <= COV_RTOL, tag
    assert np.isfinite(cov[~ok & np.isfinite(r_cov).all((1, 2))]).all(), tag
    np.testing.assert_allclose(tr, r_tr[:, 0], rtol=1e-6, err_msg=tag)


@pytest.mark.parametrize('planar', [True, False])
@pytest.mark.parametrize('wpo', [0, 1, 2, 3, 4, 8])
def test_config2_batch_matches_oracle(dev, orc, planar, wpo):
    """Seeded config-2 objects: mask/K0 bit-exact, pose 1e-4, cov 1e-5 — both layouts the pipeline
    produces (numpy pairwise vs sequential istd mean) and every wavefronts-per-object variant."""
    b = syn.make_batch(B=96, seed=1234)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=planar)
    ref = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, num_threads=0)
    gpu = _run(dev, x2d, istd, x3d, K, ur, vr, thr, flags=wpo << 8)
    _cmp(gpu, ref, f'planar={planar} wpo={wpo}')
    assert (gpu[5][:, 2] < 16).all(), 'no object of a config-2 batch is ill-conditioned (MR_DIAG_WHY_ILL_CONDITIONED)'


def test_given_init_pose_and_no_ransac(dev, orc, batch64):
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=False)
    init = np.concatenate([batch64['gt_yaw'][:, None], batch64['gt_t']], 1) + np.array([0.1, 0.3, -0.1, 1.0])
    ref = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, None, True, init_pose=init, return_diag=True)
    _cmp(_run(dev, x2d, istd, x3d, K, ur, vr, None, init=init), ref, 'given init')
    ref = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, None, False, return_diag=True)      # no RANSAC, all points in LM
    _cmp(_run(dev, x2d, istd, x3d, K, ur, vr, None, inlier_opt_only=False), ref, 'no ransac, all points')


def test_per_object_cameras_and_ranges(dev, orc, batch64):
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=True)
    n = 16
    Kb = np.repeat(K, n, 0).copy(); Kb[:, 0, 0] *= np.linspace(0.98, 1.02, n); Kb[:, 1, 2] += np.linspace(-3, 3, n)
    urb = np.repeat(ur, n, 0).copy(); urb[:, 0] -= np.arange(n)
    vrb = np.repeat(vr, n, 0).copy(); vrb[:, 1] += np.arange(n)
    ref = orc.u2d_pnp(x2d[:n], istd[:n], x3d[:n], Kb, urb, vrb, 0.5, 0.6, thr[:n], True, return_diag=True)
    _cmp(_run(dev, x2d[:n], istd[:n], x3d[:n], Kb, urb, vrb, thr[:n]), ref, 'per-object K / ranges')


def test_golden_covariance_from_the_reference(dev, g12):
    """R2/R6/R7 against the reference's own jacobian.py/hessian.py/torch.inverse outputs: the kernel's
    covariance at a GIVEN pose (LM disabled by feeding the pose as init and masking through the
    candidate set is not possible, so use zero weights outside the inlier set and 0 LM movement)."""
    from monorun_amd import _lib
    B = 6
    # make LM a no-op: evaluate at the golden pose by giving it as init with every weight intact is not a
    # no-op, so instead check the covariance stage through its own entry: COV at the init pose with
    # MR_COV_* needs the pose the LM returns.  We therefore compare on the LM's own output pose:
    x2d, istd, x3d = g12['x2d'][:B].astype(np.float32), g12['istd'][:B].astype(np.float32), g12['x3d'][:B].astype(np.float32)
    K, ur, vr = g12['K'][:B].astype(np.float32), g12['u_range'][:B].astype(np.float32), g12['v_range'][:B].astype(np.float32)
    init = np.concatenate([g12['yaw'][:B, None], g12['t'][:B]], 1)
    valid, pose, cov, tr, mask, diag = _run(dev, x2d, istd, x3d, K, ur, vr, None, init=init, thres=0.0, flags=_lib.MR_NO_ISTD_MASK)
    # recompute with the reference-pinned oracle at the SAME float32 pose
    from oracle import oracle as orc
    for b in range(B):
        _, _, H = orc.torch_jacobian(K[b], 0.5, ur[b], vr[b], pose[b, 0], pose[b, 1:], x2d[b], x3d[b], istd[b], mask[b])
        ok, c = orc.pose_cov(H)
        assert ok == valid[b]
        assert np.abs(cov[b] - c).max() / np.abs(c).max() <= COV_RTOL, b


def test_clamps_zclip_uclip_and_singular_object(dev, orc, g12):
    """Objects 1 (z-clamped points), 2 (u beyond the +-200 border), 5 (skewed K), 6 (everything clamped:
    singular J^T J -> cov = I, valid = False) from the golden fixture, solved end to end."""
    x2d, istd, x3d = g12['x2d'].astype(np.float32), g12['istd'].astype(np.float32), g12['x3d'].astype(np.float32)
    K, ur, vr = g12['K'].astype(np.float32), g12['u_range'].astype(np.float32), g12['v_range'].astype(np.float32)
    init = np.concatenate([g12['yaw'][:, None], g12['t']], 1)
    ref = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, None, True, init_pose=init, return_diag=True)
    gpu = _run(dev, x2d, istd, x3d, K, ur, vr, None, init=init)
    _cmp(gpu, ref, 'clamp fixture')
    assert not gpu[0][6] and np.array_equal(gpu[2][6], np.eye(4, dtype=np.float32))


@pytest.mark.parametrize('P', [4, 5, 63, 64, 65, 100, 784, 1000])
def test_ragged_point_counts(dev, orc, P):
    rng = np.random.default_rng(P)
    B = 8
    c = syn.cube_config1(n_points=P * B, seed=P)
    x2d = (c['pts2d'] + rng.normal(0, 0.5, c['pts2d'].shape)).reshape(B, P, 2).astype(np.float32)
    x3d = c['pts3d'].reshape(B, P, 3).astype(np.float32)
    istd = (np.exp(-rng.normal(np.log(2.0), 0.5, (B, P, 2))) / 10).astype(np.float32)
    K = c['K'][None].astype(np.float32)
    ur, vr = np.array([[-200, 1442]], np.float32), np.array([[-200, 575]], np.float32)
    thr = np.full(B, 6.0, np.float32)
    for planar in (False, True):
        if planar:
            x2d_, istd_, x3d_ = [np.ascontiguousarray(a.transpose(0, 2, 1)).transpose(0, 2, 1) for a in (x2d, istd, x3d)]
        else:
            x2d_, istd_, x3d_ = x2d, istd, x3d
        ref = orc.u2d_pnp(x2d_, istd_, x3d_, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True)
        _cmp(_run(dev, x2d_, istd_, x3d_, K, ur, vr, thr), ref, f'P={P} planar={planar}')


def test_failure_paths_match(dev, orc, batch64):
    x2d, istd, x3d, K, ur, vr, thr = [np.array(a) for a in syn.pnp_boundary(batch64, planar=False)]
    n = 6
    x2d, istd, x3d, thr = x2d[:n].copy(), istd[:n].copy(), x3d[:n].copy(), thr[:n].copy()
    thr[0] = 1e-5                       # no hypothesis reaches 5 consensus points -> initialiser fails
    x3d[1, :, :] = 1.0                  # degenerate geometry
    istd[2, 4:, :] *= 1e-3              # <= 4 istd inliers -> all points, mask all-True
    x3d[3, 7, 0] = np.nan               # NaN correspondence
    istd[4] = 0.0                       # zero weights everywhere
    ref = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True)
    gpu = _run(dev, x2d, istd, x3d, K, ur, vr, thr)
    assert not ref[0][0] and not ref[0][1]
    valid, pose, cov, tr, mask, diag = gpu
    assert np.array_equal(valid, ref[0]) and np.array_equal(mask, ref[5])
    assert np.array_equal(diag[:, 2] % 16, ref[6][:, 2])         # % 16: without MR_DIAG_WHY_ILL_CONDITIONED
    ok = ref[0]
    assert np.abs(pose[ok] - np.concatenate([ref[1], ref[2]], 1)[ok]).max() <= POSE_TOL
    assert np.array_equal(pose[~ok & (diag[:, 2] % 16 == 8)], np.zeros_like(pose[~ok & (diag[:, 2] % 16 == 8)]))
    bad = ~np.isfinite(ref[3]).all((1, 2))
    assert np.array_equal(np.isfinite(cov).all((1, 2)), ~bad)
    sc = np.abs(ref[3][~bad]).reshape((~bad).sum(), -1).max(1)[:, None, None]
    assert (np.abs(cov[~bad] - ref[3][~bad]) / sc).max() <= COV_RTOL


def test_half_and_double_inputs(dev, orc, batch64):
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=True)
    n = 24
    # fp64 tensors holding float32 values: identical problem
    ref = orc.u2d_pnp(x2d[:n], istd[:n], x3d[:n], K, ur, vr, 0.5, 0.6, thr[:n], True, return_diag=True)
    g64 = _run(dev, x2d[:n].astype(np.float64), istd[:n].astype(np.float64), x3d[:n].astype(np.float64), K, ur, vr, thr[:n],
               flags=2)   # planar copy becomes contiguous (B,P,C) after astype -> force numpy's pairwise order of the original
    _cmp(g64, ref, 'fp64 storage')
    # fp16 storage (stress-config style): the oracle sees the fp16-rounded values
    h2, hw, h3 = x2d[:n].astype(np.float16), istd[:n].astype(np.float16), x3d[:n].astype(np.float16)
    ref16 = orc.u2d_pnp(h2.astype(np.float32), hw.astype(np.float32), h3.astype(np.float32), K, ur, vr, 0.5, 0.6, thr[:n], True, return_diag=True)
    _cmp(_run(dev, h2, hw, h3, K, ur, vr, thr[:n], flags=1), ref16, 'fp16 storage')


def test_legacy_per_object_c_entry_point(dev, orc, batch64):
    """`pnp_uncert` with the reference's own signature (ext.h:1-13): host fp64 buffers, one object."""
    from monorun_amd import _lib
    lib = _lib.load()
    c = syn.cube_config1()
    dp = ctypes.POINTER(ctypes.c_double)
    def call(p2, p3, w, K, init, clips, with_cov):
        p2, p3, w, K, init, clips = [np.ascontiguousarray(a, np.float64) for a in (p2, p3, w, K, init, clips)]
        val = np.zeros(1, np.int32); pose = np.zeros(4); cov = np.eye(4); tr = np.zeros(1)
        lib.pnp_uncert(p2.ctypes.data_as(dp), p3.ctypes.data_as(dp), w.ctypes.data_as(dp), K.ctypes.data_as(dp), init.ctypes.data_as(dp),
                       val.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), pose.ctypes.data_as(dp),
                       cov.ctypes.data_as(dp) if with_cov else None, tr.ctypes.data_as(dp), p2.shape[0], clips.ctypes.data_as(dp))
        return int(val[0]), pose, cov, float(tr[0])
    val, pose, cov, tr = call(c['pts2d'], c['pts3d'], c['wgt2d'], c['K'], c['init_pose'], c['clips'], True)
    r = orc.pnp_uncert(c['pts2d'], c['pts3d'], c['wgt2d'], c['K'], c['init_pose'], c['clips'], with_cov=True)
    assert val == r['val'] == 1 and np.abs(pose - r['pose']).max() < 1e-9 and np.abs(pose - c['gt_pose']).max() < 1e-6
    assert np.abs(cov - r['cov']).max() / np.abs(r['cov']).max() < 1e-8 and abs(tr - r['tr']) <= 1e-9 * r['tr']
    # noisy object, inlier subset, no covariance requested
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=False)
    sel = ~batch64['outlier'][2].ravel()
    clips = np.array([0.5, -200, 1442, -200, 575.0])
    init = np.array([batch64['gt_yaw'][2], *batch64['gt_t'][2]]) + np.array([0.1, 0.2, 0.1, 1.0])
    val, pose, cov, tr = call(x2d[2][sel], x3d[2][sel], istd[2][sel], K[0], init, clips, False)
    r = orc.pnp_uncert(x2d[2][sel], x3d[2][sel], istd[2][sel], K[0], init, clips)
    assert val == r['val'] == 1 and np.abs(pose - r['pose']).max() < 1e-8 and np.array_equal(cov, np.eye(4))


def test_torch_level_dropin_api(dev, orc, batch64):
    """PnPUncert.forward / pnp_uncert / u2d_pnp_cpu: the reference's return contracts
    (pnp_uncert.py:87, pnp_uncert_cpu.py:193-209) on device and host tensors."""
    from monorun_amd.ops import build_pnp, pnp_uncert, u2d_pnp_cpu
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=True)
    m = build_pnp(dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, forward_exact_hessian=False))
    t = lambda a: _dv(a, dev)
    ret, r_vec, t_vec, cov, mask = m(t(x2d), t(istd), t(x3d), t(K), t(ur), t(vr), t(thr))
    assert ret.dtype == torch.bool and ret.shape == (64,) and r_vec.shape == (64, 1) and t_vec.shape == (64, 3)
    assert cov.shape == (64, 4, 4) and mask.dtype == torch.bool and mask.shape == (64, 784)
    assert all(o.device.type == 'cuda' for o in (ret, r_vec, t_vec, cov, mask)) and cov.dtype == torch.float32
    ref = orc.u2d_pnp_epnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True)        # the module built from the reference's dict runs the REFERENCE's flow
    assert m.initialiser == 'epnp' and np.array_equal(mask.cpu().numpy(), ref[5]) and np.abs(t_vec.cpu().numpy() - ref[2]).max() <= POSE_TOL
    # ... and the explicit fast mode the K0 specification's
    mk = build_pnp(dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, forward_exact_hessian=False, initialiser='k0'))
    rk = mk(t(x2d), t(istd), t(x3d), t(K), t(ur), t(vr), t(thr))
    refk = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True)
    assert np.array_equal(rk[4].cpu().numpy(), refk[5]) and np.abs(rk[2].cpu().numpy() - refk[2]).max() <= POSE_TOL
    # host tensors in -> host tensors out (staged through the GPU, never solved on the CPU)
    ret_h, r_h, t_h, cov_h, mask_h = pnp_uncert(torch.from_numpy(x2d), torch.from_numpy(istd), torch.from_numpy(x3d), torch.from_numpy(K),
                                                torch.from_numpy(ur), torch.from_numpy(vr), 0.5, 0.6, torch.from_numpy(thr), True)
    assert ret_h.device.type == 'cpu' and torch.equal(t_h, t_vec.cpu()) and torch.equal(mask_h, mask.cpu())
    # numpy-level driver, with the Ceres-style covariance
    o = u2d_pnp_cpu(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, with_pose_cov=True)
    assert [a.shape for a in o] == [(64,), (64, 1), (64, 3), (64, 4, 4), (64, 1), (64, 784)]
    assert o[0].dtype == bool and o[5].dtype == bool and np.array_equal(o[5], ref[5]) and np.array_equal(o[2], t_vec.cpu().numpy())
    o2 = u2d_pnp_cpu(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, with_pose_cov=False)
    assert o2[3] is None
    # empty batch on the device
    e = m(t(x2d[:0]), t(istd[:0]), t(x3d[:0]), t(K), t(ur), t(vr), t(thr[:0]))
    assert [tuple(a.shape) for a in e] == [(0,), (0, 1), (0, 3), (0, 4, 4), (0, 784)]
    # coord_istd_normalize pre-op (pnp_uncert.py:130-132)
    mn = build_pnp(dict(type='PnPUncert', coord_istd_normalize=True))
    out_n = mn(t(x2d[:8]), t(istd[:8]), t(x3d[:8]), t(K), t(ur), t(vr), t(thr[:8]))
    assert out_n[0].all()


def test_full_size_properties(dev):
    """BASELINE config 2 at full size (1024 x 784): size-independent properties instead of the oracle."""
    b = syn.make_batch(B=1024, seed=1234)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    valid, pose, cov, tr, mask, diag = _run(dev, x2d, istd, x3d, K, ur, vr, thr)
    assert valid.mean() > 0.99
    dy = np.abs(np.angle(np.exp(1j * (pose[:, 0] - b['gt_yaw']))))
    rel_t = np.linalg.norm(pose[:, 1:] - b['gt_t'], axis=1) / np.linalg.norm(b['gt_t'], axis=1)
    assert np.median(dy[valid]) < 0.03 and np.median(rel_t[valid]) < 0.02
    # covariances are symmetric positive definite
    assert np.abs(cov - cov.transpose(0, 2, 1)).max() <= 1e-6 * np.abs(cov).max()
    assert (np.linalg.eigvalsh(cov[valid].astype(np.float64)) > 0).all()
    # idempotence + permutation equivariance: objects are independent, results do not depend on batch position
    perm = np.random.default_rng(0).permutation(1024)
    v2, p2, c2, t2, m2, d2 = _run(dev, x2d[perm], istd[perm], x3d[perm], K, ur, vr, thr[perm])
    assert np.array_equal(p2, pose[perm]) and np.array_equal(m2, mask[perm]) and np.array_equal(c2, cov[perm])
    # re-solving from the returned pose with the returned inlier set as the only candidates is a fixed point (<= one LM step away)
    from monorun_amd import _lib
    w = np.array(istd) * mask[:, :, None]
    v3, p3, c3, t3, m3, d3 = _run(dev, x2d, w, x3d, K, ur, vr, None, init=pose.astype(np.float64), flags=_lib.MR_NO_ISTD_MASK, inlier_opt_only=False)
    ok = valid & v3
    assert np.abs(p3[ok] - pose[ok]).max() < 5e-2 and (d3[ok, 0] <= 3).mean() > 0.95
    # the LM only ever lowers the cost it was given
    assert (diag[valid, 1] >= 0).all()


def test_stress_shape_fp16_56x56(dev, orc):
    """BASELINE config 5 shape (56x56 = 3136 correspondences, fp16 storage of X3d / istd / x2d) at a
    batch the oracle finishes in seconds: 88 KB fp32-equivalent tiles (44 KB as fp16) in LDS, the
    32-leaf numpy pairwise tree, 49 points per lane at one wave per object."""
    b = syn.make_batch(B=24, hw=56, seed=4321)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    # fp16 storage: pixel coordinates up to ~1442 keep 1-px resolution in fp16 — the oracle sees the same rounded values
    h2, hw_, h3 = x2d.astype(np.float16), istd.astype(np.float16), x3d.astype(np.float16)
    mk = lambda a: np.ascontiguousarray(a.transpose(0, 2, 1)).transpose(0, 2, 1)      # keep the planar strides after astype
    h2, hw_, h3 = mk(h2), mk(hw_), mk(h3)
    assert h2.strides == (2 * 3136 * 2, 2, 3136 * 2)
    ref = orc.u2d_pnp(mk(h2.astype(np.float32)), mk(hw_.astype(np.float32)), mk(h3.astype(np.float32)), K, ur, vr, 0.5, 0.6, thr, True,
                      return_diag=True, num_threads=0)
    for wpo in (1, 2, 4):
        _cmp(_run(dev, h2, hw_, h3, K, ur, vr, thr, flags=wpo << 8), ref, f'fp16 56x56 wpo={wpo}')
    # fp32 storage of the same shape
    ref32 = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, num_threads=0)
    _cmp(_run(dev, x2d, istd, x3d, K, ur, vr, thr), ref32, 'fp32 56x56')


@pytest.mark.parametrize('seed,B', [(1234, 1024), (1, 256), (2, 256), (20260928, 256)])
def test_bench_workload_matches_oracle_object_by_object(dev, orc, seed, B):
    """The exact bench.py workload (config 2, seed 1234, 1024 objects, planar fp32) and three more seeds:
    every object against the oracle (mask/K0 bit-exact, pose 1e-4, cov 1e-5, same LM iteration counts)."""
    b = syn.make_batch(B=B, seed=seed)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    ref = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, num_threads=0)
    _cmp(_run(dev, x2d, istd, x3d, K, ur, vr, thr), ref, f'seed={seed} B={B}')


def test_side_stream_misaligned_and_strided_inputs(dev, orc, batch64):
    """(a) launch on a non-default stream (asynchronous, results valid after that stream syncs);
    (b) planar input whose base is only 4-byte aligned -> the dword LDS-DMA path;
    (c) non-contiguous object blocks (every second point of a larger tensor) -> the gather path;
    (d) fp16 planar input with odd byte counts -> the element-wise copy path."""
    from monorun_amd.ops.least_squares.pnp_uncert import PnPLaunch
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=True)
    n = 32
    ref = orc.u2d_pnp(x2d[:n], istd[:n], x3d[:n], K, ur, vr, 0.5, 0.6, thr[:n], True, return_diag=True)
    t = lambda a: _dv(a, dev)
    # (a) side stream
    s = torch.cuda.Stream(device=dev)
    L = PnPLaunch(t(x2d[:n]), t(istd[:n]), t(x3d[:n]), t(K), t(ur), t(vr), 0.5, 0.6, t(thr[:n]), True, with_diag=True)
    L.run(s.cuda_stream)
    s.synchronize()
    _cmp((L.valid.cpu().numpy().astype(bool), L.pose.cpu().numpy(), L.cov.cpu().numpy(), L.tr.cpu().numpy(),
          L.mask.cpu().numpy().astype(bool), L.diag.cpu().numpy()), ref, 'side stream')

    # (b) 4-byte aligned base: allocate one extra float in front and view from element 1
    def shifted(a):
        flat = torch.empty(a.size + 1, dtype=torch.float32, device=dev)
        base = np.ascontiguousarray(np.asarray(a).transpose(0, 2, 1))            # (n, C, P) contiguous = planar memory image
        flat[1:] = torch.from_numpy(base.reshape(-1)).to(dev)
        nb, C, P = base.shape
        return torch.as_strided(flat, (nb, P, C), (C * P, 1, P), storage_offset=1)
    xs, ws_, x3s = shifted(x2d[:n]), shifted(istd[:n]), shifted(x3d[:n])
    assert xs.data_ptr() % 16 == 4
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
    out = pnp_uncert_device(xs, ws_, x3s, t(K), t(ur), t(vr), 0.5, 0.6, t(thr[:n]), True, with_diag=True)
    torch.cuda.synchronize()
    o = [v.cpu().numpy() for v in out]
    _cmp((o[0].astype(bool), o[1], o[2], o[3], o[4].astype(bool), o[5]), ref, 'misaligned planar')

    # (c) every second point of a (n, 2P, C) contiguous tensor: object blocks are not contiguous
    def interleave(a):
        a = np.ascontiguousarray(a)
        big = np.zeros((a.shape[0], 2 * a.shape[1], a.shape[2]), np.float32)
        big[:, ::2] = a
        return t(big)[:, ::2]
    xi, wi, x3i = interleave(x2d[:n]), interleave(istd[:n]), interleave(x3d[:n])
    assert not xi.is_contiguous()
    refc = orc.u2d_pnp(np.ascontiguousarray(x2d[:n]), np.ascontiguousarray(istd[:n]), np.ascontiguousarray(x3d[:n]), K, ur, vr,
                       0.5, 0.6, thr[:n], True, return_diag=True)                 # stride_p != 1 -> numpy's sequential mean order
    out = pnp_uncert_device(xi, wi, x3i, t(K), t(ur), t(vr), 0.5, 0.6, t(thr[:n]), True, with_diag=True)
    torch.cuda.synchronize()
    o = [v.cpu().numpy() for v in out]
    _cmp((o[0].astype(bool), o[1], o[2], o[3], o[4].astype(bool), o[5]), refc, 'strided gather')

    # (d) fp16, P = 101 (C*P*2 bytes not a multiple of 4 for C = 3)
    rng = np.random.default_rng(9)
    P = 101
    c = syn.cube_config1(n_points=P * 8, seed=4)
    h2 = (c['pts2d'] + rng.normal(0, 0.5, c['pts2d'].shape)).reshape(8, P, 2).astype(np.float16)
    h3 = c['pts3d'].reshape(8, P, 3).astype(np.float16)
    hw = (np.exp(-rng.normal(np.log(2.0), 0.5, (8, P, 2))) / 10).astype(np.float16)
    pl = lambda a: np.ascontiguousarray(a.transpose(0, 2, 1)).transpose(0, 2, 1)
    h2, h3, hw = pl(h2), pl(h3), pl(hw)
    Kc = c['K'][None].astype(np.float32)
    thr8 = np.full(8, 6.0, np.float32)
    ref16 = orc.u2d_pnp(pl(h2.astype(np.float32)), pl(hw.astype(np.float32)), pl(h3.astype(np.float32)), Kc, ur, vr, 0.5, 0.6, thr8, True,
                        return_diag=True)
    _cmp(_run(dev, h2, hw, h3, Kc, ur, vr, thr8), ref16, 'fp16 odd sizes')


@pytest.mark.gpu
def test_adversarial_inputs_terminate_and_match_oracle_validity(dev, orc):
    """Degenerate / non-finite inputs (coincident points, NaN and Inf correspondences, zero or negative weights, overflow,
    garbage geometry, zero consensus threshold): every launch terminates, no non-finite pose is reported valid, and the
    valid flags equal the oracle's — also exercises the leader/follower exits of the LM (evaluation failure, refit failure)."""
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
    rng = np.random.default_rng(11)
    for trial in range(16):
        B = int(rng.choice([1, 3, 64])); hw = int(rng.choice([2, 3, 10, 28]))
        b = syn.make_batch(B=B, hw=hw, seed=int(rng.integers(1 << 30)))
        x2d, istd, x3d, K, ur, vr, thr = [np.array(a, copy=True) for a in syn.pnp_boundary(b, planar=bool(rng.integers(2)))]
        P = x2d.shape[1]
        sel = rng.uniform(size=B) < 0.5
        mode = trial % 8
        if mode == 0: x3d[sel] = 0.0
        elif mode == 1: x2d[sel, rng.integers(P)] = np.nan
        elif mode == 2: istd[sel] = 0.0
        elif mode == 3: x3d[sel] *= 1e20
        elif mode == 4: istd[sel] = -istd[sel]
        elif mode == 5: x3d[sel] = rng.normal(0, 1, x3d[sel].shape).astype(np.float32)
        elif mode == 6: thr[sel] = 0.0
        else: x2d[sel] = np.inf
        with np.errstate(all='ignore'):
            ref = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, num_threads=0)
        for wpo in (0, 1, 4):
            out = pnp_uncert_device(*[_dv(a, dev) for a in (x2d, istd, x3d, K, ur, vr)], 0.5, 0.6, _dv(thr, dev), True, flags=(wpo << 8))
            torch.cuda.synchronize()
            valid, pose = out[0].cpu().numpy().astype(bool), out[1].cpu().numpy()
            assert np.isfinite(pose[valid]).all(), (trial, mode, wpo)
            assert np.array_equal(valid, ref[0]), (trial, mode, wpo)


def test_wave_counts_give_bitwise_equal_results(dev):
    """the one- and two-wave instantiations evaluate the LM point loop and the covariance pass first as if no clamp of the
    functor were active (and redo a lane's sums with the full form when one was); the four-wave kernel has the full form only.  Where
    no clamp is active both forms must give the SAME bits — a property that rests on the compiler contracting the two forms alike, so
    it is pinned here: validity, poses, covariances and masks of 1, 2, 4 and 8 waves per object are bitwise equal, the LM's iteration
    counts and exit reasons identical, on ordinary objects, on objects whose projections leave the u / v range (clamps active) and
    for the from-init (EXT) launch.  (The final trust-region radius and cost are float32 roundings of quantities that depend on the
    summation order over the waves through the step-quality ratio: equal to 1e-6, one float32 ulp apart on 1 of 512 objects.)"""
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device, epnp_ransac_device, pnp_uncert_from_init_device
    b = syn.make_batch(B=512, seed=77)
    x2d, istd, x3d, K, ur, vr, thr = [_dv(a, dev) for a in syn.pnp_boundary(b, planar=True)]
    tight_u = torch.tensor([[300.0, 900.0]], device=dev); tight_v = torch.tensor([[120.0, 260.0]], device=dev)      # many points get clamped

    def same(a, r, what):
        valid, pose, cov, tr, mask, diag = a
        assert torch.equal(valid, r[0]) and torch.equal(pose, r[1]) and torch.equal(cov, r[2]) and torch.equal(mask, r[4]), what
        assert torch.equal(diag[:, [0, 2, 3]], r[5][:, [0, 2, 3]]), what
        torch.testing.assert_close(tr, r[3], rtol=1e-6, atol=0); torch.testing.assert_close(diag[:, 1], r[5][:, 1], rtol=1e-6, atol=0)
    for rng_u, rng_v in ((ur, vr), (tight_u, tight_v)):
        outs = {w: pnp_uncert_device(x2d, istd, x3d, K, rng_u, rng_v, 0.5, 0.6, thr, True, flags=w << 8, with_diag=True) for w in (1, 2, 4, 8)}
        torch.cuda.synchronize()
        for w in (1, 2, 8):
            same(outs[w], outs[4], f'{w} waves per object differ from 4')
        assert int(outs[4][0].sum()) > 400
    ini, im, iv, _, _ = epnp_ransac_device(x2d, istd, x3d, K, epnp_istd_thres=0.6, epnp_ransac_thres=thr)
    for rng_u, rng_v in ((ur, vr), (tight_u, tight_v)):
        outs = {w: pnp_uncert_from_init_device(x2d, istd, x3d, K, rng_u, rng_v, ini, im, iv, z_min=0.5, inlier_opt_only=True, flags=w << 8, with_diag=True) for w in (1, 2, 4)}
        torch.cuda.synchronize()
        for w in (1, 2):
            same(outs[w], outs[4], f'from-init launch: {w} waves per object differ from 4')


def test_hip_kernel_against_the_reference_initialiser_restated(dev, orc):
    """R5: the explicit FAST MODE (initialiser='k0', the one-launch kernel; not the default since round 5) against the reference's flow with its OWN initialiser restated (EPnP inside
    OpenCV's RANSAC loop, oracle/epnp.inc) on a config-2 batch, compared after the LM: same validity, the same inlier set
    for the large majority of objects, poses within a small fraction of the posterior standard deviation
    (distribution: DESIGN.md §5; CPU counterpart: tests/test_oracle_epnp.py)."""
    from test_oracle_epnp import compare_runs
    b = syn.make_batch(B=512, seed=1234)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    valid, pose, cov, tr, mask, diag = _run(dev, x2d, istd, x3d, K, ur, vr, thr)
    c = [np.ascontiguousarray(a) for a in (x2d, istd, x3d)]
    ref = orc.u2d_pnp_epnp(c[0], c[1], c[2], K, ur, vr, 0.5, 0.6, thr, True, num_threads=0)
    s = compare_runs((valid, pose[:, :1], pose[:, 1:], cov, tr, mask), ref)
    assert s['valid0'].all() and s['valid1'].all()
    assert s['same'].mean() >= 0.80 and s['iou'].mean() >= 0.995 and s['iou'].min() >= 0.80
    sm, df = s['same'] & s['ok'], ~s['same'] & s['ok']
    assert np.nanquantile(s['maha'][sm], 0.99) <= 0.05 and np.mean(s['maha'][sm] > 0.1) <= 0.01
    assert np.nanmax(s["maha"][df]) <= 5.0 and np.nanquantile(s['maha'][df], 0.5) <= 0.1


def test_config5_full_size_on_one_gpu(dev, orc):
    """BASELINE config 5 at its FULL size — 65 536 proposals x 56x56 correspondences, fp16 storage (2.9 GB of inputs) — in one
    launch on one GPU (the 8-GPU run shards it 8 192 per rank; a single MI355X holds it 100 times over): 256 distinct objects
    tiled 256x, so that (a) the first tile is compared with the oracle object by object and (b) every other tile must
    reproduce it bit for bit (results do not depend on batch position or on what else is in flight)."""
    nd, rep, hw = 256, 256, 56
    b = syn.make_batch(B=nd, hw=hw, seed=4321)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    mk = lambda a: np.ascontiguousarray(a.astype(np.float16).transpose(0, 2, 1))                     # (nd, C, P) fp16, channel-planar
    t = lambda a: torch.from_numpy(a).to(dev)
    big = [t(mk(a)).repeat(rep, 1, 1).permute(0, 2, 1) for a in (x2d, istd, x3d)]                    # (65536, P, C) views, strides (C*P, 1, P)
    assert big[0].shape == (nd * rep, hw * hw, 2) and big[0].stride() == (2 * hw * hw, 1, hw * hw)
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
    valid, pose, cov, tr, mask, diag = pnp_uncert_device(big[0], big[1], big[2], t(np.asarray(K)), t(np.asarray(ur)), t(np.asarray(vr)), 0.5, 0.6,
                                                         t(np.asarray(thr)).repeat(rep), True, with_diag=True)
    torch.cuda.synchronize()
    f32 = lambda a: np.ascontiguousarray(mk(a).astype(np.float32).transpose(0, 2, 1))
    ref = orc.u2d_pnp(f32(x2d), f32(istd), f32(x3d), K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, num_threads=0)
    first = [v[:nd].cpu().numpy() for v in (valid, pose, cov, tr, mask, diag)]
    _cmp((first[0].astype(bool), first[1], first[2], first[3], first[4].astype(bool), first[5]), ref, 'config 5, first tile')
    for v in (valid, pose, cov, tr, mask):
        tiles = v.view(rep, nd, *v.shape[1:])
        assert bool((tiles == tiles[:1]).all()), 'a later tile differs from the first'
    assert float(valid.float().mean()) > 0.97


@pytest.mark.gpu
def test_half_a_million_objects_in_one_launch_64bit_offsets(dev, orc):
    """500 000 objects x 28x28 in ONE launch: 11 GB of correspondences, the X3d tensor alone is 4.7 GB — every per-object base offset
    beyond 2^32 bytes, grid of 500 000 workgroups.  500 distinct objects tiled 1000x: the first tile is compared with the oracle,
    every other tile must reproduce it bit for bit."""
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
    nd, rep = 500, 1000
    b = syn.make_batch(B=nd, seed=77)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(dev)
    planar = lambda a: t(np.ascontiguousarray(np.asarray(a).transpose(0, 2, 1))).repeat(rep, 1, 1).permute(0, 2, 1)      # (B, P, C) views of (B, C, P)
    big = [planar(a) for a in (x2d, istd, x3d)]
    assert big[2].numel() * 4 > 2 ** 32 and big[0].shape[0] == nd * rep
    valid, pose, cov, tr, mask, diag = pnp_uncert_device(big[0], big[1], big[2], t(K), t(ur), t(vr), 0.5, 0.6, t(thr).repeat(rep), True, with_diag=True)
    torch.cuda.synchronize()
    ref = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, num_threads=0)
    first = [v[:nd].cpu().numpy() for v in (valid, pose, cov, tr, mask, diag)]
    _cmp((first[0].astype(bool), first[1], first[2], first[3], first[4].astype(bool), first[5]), ref, '500k objects, first tile')
    for v in (valid, pose, cov, tr, mask):
        tiles = v.view(rep, nd, *v.shape[1:])
        assert bool((tiles == tiles[:1]).all()), 'a later tile differs from the first'


@pytest.mark.gpu
def test_legacy_entry_point_from_many_threads(dev, orc, batch64):
    """The reference's per-object symbol is documented as stateless and re-entrant (ext.h:1-13; SURVEY 8b "Threading"): eight host
    threads call `pnp_uncert` concurrently (ctypes releases the GIL; the library serialises on its staging buffers) on different
    objects — every result equals the serial one bit for bit, while batched launches run on another stream in between."""
    import threading
    from monorun_amd import _lib
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
    lib = _lib.load()
    dp = ctypes.POINTER(ctypes.c_double)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=False)
    clips = np.array([0.5, -200, 1442, -200, 575.0])
    jobs = []
    for o in range(32):
        sel = ~batch64['outlier'][o].ravel()
        init = np.array([batch64['gt_yaw'][o], *batch64['gt_t'][o]]) + np.array([0.1, 0.2, 0.1, 1.0])
        jobs.append([np.ascontiguousarray(a, np.float64) for a in (x2d[o][sel], x3d[o][sel], istd[o][sel], K[0], init, clips)])

    def call(j):
        p2, p3, w, Kd, init, cl = j
        val = np.zeros(1, np.int32); pose = np.zeros(4); cov = np.eye(4); tr = np.zeros(1)
        lib.pnp_uncert(p2.ctypes.data_as(dp), p3.ctypes.data_as(dp), w.ctypes.data_as(dp), Kd.ctypes.data_as(dp), init.ctypes.data_as(dp),
                       val.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), pose.ctypes.data_as(dp), cov.ctypes.data_as(dp), tr.ctypes.data_as(dp),
                       p2.shape[0], cl.ctypes.data_as(dp))
        return int(val[0]), pose, cov, float(tr[0])
    serial = [call(j) for j in jobs]
    out = [None] * len(jobs)

    def worker(k):
        for rep in range(3):
            for i in range(k, len(jobs), 8):
                out[i] = call(jobs[i])
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    args = [t(a) for a in (x2d, istd, x3d, K, ur, vr)]
    side = torch.cuda.Stream(device=dev)
    th = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for x in th:
        x.start()
    with torch.cuda.stream(side):
        batched = [pnp_uncert_device(*args, 0.5, 0.6, t(thr), True) for _ in range(20)]
    for x in th:
        x.join()
    torch.cuda.synchronize()
    for (v0, p0, c0, t0), (v1, p1, c1, t1) in zip(serial, out):
        assert v0 == v1 == 1 and np.array_equal(p0, p1) and np.array_equal(c0, c1) and t0 == t1
    for b in batched[1:]:
        assert torch.equal(b[1], batched[0][1]) and torch.equal(b[4], batched[0][4])
    assert int(batched[0][0].sum()) >= 60


def test_reference_eigenvalue_rule_for_ill_conditioned_hessians(dev, orc):
    """pnp_uncert.py:77-85: in the branch the reference takes when torch.inverse raises, an object stays valid only if
    lambda_min(h) > max(1e-6 lambda_max(h), 0), the others get h := I.  The fused kernel alone drops an object only when h has no
    Cholesky factorisation; `mr_cov_symeig_rule` / pnp_uncert(..., cov_symeig_rule=True) applies the eigenvalue test per object.
    (a) the pass against numpy on synthetic covariances spanning condition numbers 1 ... 1e12, indefinite, NaN;
    (b) objects whose yaw is unobservable (all correspondences on the rotation axis ... nearly): valid without the rule, dropped with it."""
    from monorun_amd.ops.least_squares.pnp_uncert import cov_symeig_rule_device
    from monorun_amd.ops import pnp_uncert
    rng = np.random.default_rng(12)
    covs = []
    for cond in (1.0, 1e2, 1e5, 9e5, 1.1e6, 1e7, 1e12):
        for _ in range(8):
            q, _ = np.linalg.qr(rng.normal(size=(4, 4)))
            lam = np.array([1.0, cond ** (1 / 3), cond ** (2 / 3), cond]) * rng.uniform(1e-4, 1e2)
            covs.append(q @ np.diag(lam) @ q.T)
    q, _ = np.linalg.qr(rng.normal(size=(4, 4)))
    covs.append(q @ np.diag([-1e-3, 1.0, 2.0, 3.0]) @ q.T)                   # indefinite
    covs.append(np.full((4, 4), np.nan)); covs.append(np.eye(4))
    cov = np.stack(covs).astype(np.float32)
    valid = np.ones(len(cov), np.uint8)
    dv, dc = torch.from_numpy(valid).to(dev), torch.from_numpy(cov).to(dev)
    lam = cov_symeig_rule_device(dv, dc, with_eigs=True)
    torch.cuda.synchronize()
    r_valid, r_cov, r_lam = orc.cov_symeig_rule(valid, cov)
    # the eigenvalue test itself is compared away from its own threshold (float32 covariances: the ratio is known to ~1e-6 relative)
    ratio = r_lam[:, 0] / np.where(r_lam[:, 1] != 0, r_lam[:, 1], 1.0)
    clear = ~np.isfinite(ratio) | (np.abs(ratio / 1e-6 - 1.0) > 0.05)
    assert np.array_equal(dv.cpu().numpy().astype(bool)[clear], r_valid[clear])
    same = dv.cpu().numpy().astype(bool) == r_valid
    assert np.array_equal(dc.cpu().numpy()[same], r_cov[same])
    fin = np.isfinite(r_lam).all(1)
    assert np.allclose(lam.cpu().numpy()[fin], r_lam[fin], rtol=1e-4, atol=1e-7 * np.abs(r_lam[fin]).max(1, keepdims=True))
    assert r_valid[:32].all() and not r_valid[40:56].any()                      # cond <= 9e5 kept, >= 1e7 dropped
    # (b) end to end: a far object seen through correspondences that all lie within 2 mm of the rotation axis: the yaw column of J is ~0
    b = syn.make_batch(B=24, seed=5)
    x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
    x3d = x3d.copy()
    x3d[:6, :, 0] *= 1e-3; x3d[:6, :, 2] *= 1e-3
    t = lambda a: torch.from_numpy(a).to(dev)
    init = np.concatenate([b['gt_yaw'][:, None], b['gt_t']], 1)
    outs = {}
    for rule in (False, True):
        outs[rule] = [o.cpu().numpy() for o in pnp_uncert(t(x2d), t(istd), t(x3d), t(K), t(ur), t(vr), 0.5, 0.6, None, False, cov_symeig_rule=rule, initialiser='k0')]
    h_valid, h_cov, h_lam = orc.cov_symeig_rule(outs[False][0], outs[False][3])
    clear = np.abs((h_lam[:, 0] / h_lam[:, 1]) / 1e-6 - 1.0) > 0.05
    assert np.array_equal(outs[True][0][clear], h_valid[clear])
    assert outs[False][0][:6].all() and not outs[True][0][:6].any(), 'the six degenerate objects are valid without the rule and dropped by it'
    assert np.array_equal(outs[True][3][~outs[True][0]], np.broadcast_to(np.eye(4, dtype=np.float32), (int((~outs[True][0]).sum()), 4, 4)))
    assert outs[True][0][6:].sum() >= 16


def test_bench_line_describes_the_regime_it_measured():
    """bench.py's ONE JSON line: `value` is the REFERENCE's flow (what the drop-in boundary runs) with launch sets
    in flight, the top-level roofline is the timed regime's (chip-level bytes / wall) with the longest launch under `dominant_kernel`,
    `reference_flow` carries one call at a time and the launch split, the explicit fast mode rides under `k0_fast_mode` (a child run of
    `--flow k0`, whose own line keeps rounds 1-4's layout)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '8', '--warmup', '2', '--batches', '4', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=1500)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-500:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d['metric'].startswith('PnP solves/sec') and d['unit'] == 'solves/s' and d['dtype'] == 'f64' and d['n_gpus'] == 1 and d['steps'] == 8
    assert d['config']['flow'].startswith('reference') and 1 <= d['config']['calls_per_launch_set'] <= 8 and 'solvePnPRansac' in d['config']['stages']
    rf = d['roofline']
    assert rf['bound'] == 'hbm' and rf['peak'] == 8000.0 and rf['kernel'].startswith('reference flow, 6 launches per call of fewer than 2048 objects') and '7 per launch set' in rf['kernel'] and rf['launches_in_flight'] >= 1
    chip = d['value'] * rf['algorithmic_bytes_per_launch'] / 1024 / 1e9                   # algorithmic bytes of all calls / wall time
    assert abs(rf['achieved'] - chip) <= 1e-6 * chip and abs(rf['frac'] - chip / 8000.0) <= 1e-9
    dk = rf['dominant_kernel']
    assert dk['kernel'].startswith('pnp_uncert_refit_kernel<float, ') and dk['avg_launch_ms'] > 0
    assert abs(dk['frac'] - rf['algorithmic_bytes_per_launch'] / (dk['avg_launch_ms'] * 1e-3) / 1e9 / 8000.0) <= 1e-9
    iso = rf['isolated_launch']
    assert iso['kernel_ms_avg'] > dk['avg_launch_ms'] and abs(iso['frac'] - rf['algorithmic_bytes_per_launch'] / (iso['kernel_ms_avg'] * 1e-3) / 1e9 / 8000.0) <= 1e-9
    assert d['steady_state']['steps'] >= 240 and d['steady_state']['value'] > 0 and d['single_stream']['value'] > 0
    ref = d['reference_flow']
    assert ref['in_flight']['value'] == d['value'] and ref['one_call_at_a_time']['value'] == d['single_stream']['value'] > 0
    sp = ref['launch_split']
    assert sp['initialiser_launches_ms'] > sp['lm_launch_ms'] > 0 and abs(sp['initialiser_launches_ms'] + sp['lm_launch_ms'] - iso['kernel_ms_avg']) < 0.2 * iso['kernel_ms_avg']
    assert d['outputs_verified'] is True and d['valid_fraction'] > 0.95
    pi = ref['per_image_B100']                                 # the regime the pipeline runs (one image, 100 proposals) through the DEFAULT flow
    assert pi['objects'] == 100 and pi['valid'] >= 95 and pi['outputs_of_the_three_paths_equal'] is True and pi['gpu_us_per_call_hip_events'] > 0
    for row in ('eager_pose_from_head', 'prepared_launch', 'hip_graph_replay'):
        assert pi[row]['wall_us_per_call_synced'] > 0 and pi[row]['issue_us_per_call'] > 0
    assert '6 launches' in iso['kernel'] and (iso['valu_issue'] is None or iso['valu_issue']['valu_insts_per_call'] > 1e6)
    pw = d['config']['prewarm']                                # the untimed pre-conditioning is reported, with the window as a cold process sees it
    assert pw['launches'] > 0 and pw['launches_asked'] > 0 and pw['ms'] > 0 and pw['window_before']['steps'] == 8 and pw['window_before']['value'] > 0
    k0 = d['k0_fast_mode']                                     # the child run's line, condensed
    assert k0['value'] > d['value'] and k0['steady_state']['value'] > 0 and k0['single_stream']['value'] > 0
    assert k0['roofline']['kernel'].startswith('pnp_uncert_kernel<float, ') and k0['roofline']['kernel'].endswith('false>')
    assert 'head_to_pose_1024' in k0['secondary_throughput'] and 'epnp_initialiser' not in k0['secondary_throughput']

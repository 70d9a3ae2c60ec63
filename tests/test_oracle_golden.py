"""Pin the oracle against the golden vectors produced by the reference's own importable code
(tests/golden/make_golden.py): R2/R7/R6 (jacobian.py, hessian.py, torch.inverse), R8-R13 (coders,
slice_pred, pose-head prep).  CPU only."""
import numpy as np
import pytest


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_torch_jacobian_matches_reference_fp64(orc, g12):
    B = g12['x2d'].shape[0]
    for b in range(B):
        jac, err, H = orc.torch_jacobian(g12['K'][b], float(g12['z_min']), g12['u_range'][b], g12['v_range'][b],
                                         g12['yaw'][b], g12['t'][b], g12['x2d'][b], g12['x3d'][b], g12['istd'][b], g12['inlier'][b])
        ref_jac = np.concatenate([g12['jac_yaw_f64'][b], g12['jac_t_f64'][b]], axis=2)      # (P,2,4) [yaw,tx,ty,tz]
        assert np.abs(jac - ref_jac).max() <= 1e-12 * max(1.0, np.abs(ref_jac).max()), b
        assert np.abs(err - g12['err_f64'][b]).max() <= 1e-10, b
        assert _rel(H, g12['h_f64'][b]) <= 1e-12, b
        _, _, Hn = orc.torch_jacobian(g12['K'][b], float(g12['z_min']), g12['u_range'][b], g12['v_range'][b],
                                      g12['yaw'][b], g12['t'][b], g12['x2d'][b], g12['x3d'][b], g12['istd'][b], None)
        assert _rel(Hn, g12['h_nomask_f64'][b]) <= 1e-12, b


def test_special_cases_are_exercised(g12):
    assert g12['n_zclip'][1] > 100 and g12['n_uclip'][2] > 100 and g12['n_uclip'][6] == 784
    assert 0 < g12['inlier'][3].sum() < 784
    assert g12['K'][5, 0, 1] != 0 and g12['K'][5, 1, 0] != 0            # skewed K exercises the full-K path


def test_pose_cov_matches_torch_inverse(orc, g12):
    for b in range(6):
        ok, cov = orc.pose_cov(g12['h_f64'][b])
        assert ok
        assert _rel(cov, g12['cov_f64'][b]) <= 1e-9, b
        # the reference's own fp32 pipeline agrees with its fp64 one only to fp32 accuracy
        assert _rel(cov, g12['cov_f32'][b]) <= 5e-3, b
    ok, cov = orc.pose_cov(g12['h_f64'][6])                             # singular J^T J -> identity + invalid
    assert not ok and np.array_equal(cov, np.eye(4))


def test_broadcast_camera_and_ranges(orc, g12):
    for b in range(5):
        _, _, H = orc.torch_jacobian(g12['K'][0], float(g12['z_min']), g12['u_range'][0], g12['v_range'][0],
                                     g12['yaw'][b], g12['t'][b], g12['x2d'][b], g12['x3d'][b], g12['istd'][b], g12['inlier'][b])
        assert _rel(H, g12['h_bcast_f64'][b]) <= 1e-12


def test_ceres_jacobian_equals_torch_jacobian_where_nothing_clips(orc, g12):
    """R1 (autodiff semantics) and R2 (analytic torch) are the same function away from the clamps."""
    for b in (0, 3, 4):                       # objects without z/u/v clipping and unskewed K
        clips = [float(g12['z_min']), *g12['u_range'][b], *g12['v_range'][b]]
        pose = np.array([g12['yaw'][b], *g12['t'][b]])
        res, jac = orc.residual_jacobian(g12['K'][b], clips, pose, g12['x2d'][b], g12['x3d'][b], g12['istd'][b])
        tj, terr, _ = orc.torch_jacobian(g12['K'][b], float(g12['z_min']), g12['u_range'][b], g12['v_range'][b],
                                         g12['yaw'][b], g12['t'][b], g12['x2d'][b], g12['x3d'][b], g12['istd'][b], None)
        assert np.abs(jac - tj).max() <= 1e-10 * max(1.0, np.abs(tj).max())
        assert np.abs(res - terr).max() <= 1e-10


def test_ceres_jacobian_is_the_derivative_of_its_residual(orc, g12):
    """Finite differences of R1's residual, including the clamped objects (1: z-clip, 2: u-clip)."""
    for b in range(6):
        clips = [float(g12['z_min']), *g12['u_range'][b], *g12['v_range'][b]]
        pose = np.array([g12['yaw'][b], *g12['t'][b]])
        Kb = g12['K'][b].copy(); Kb[0, 1] = Kb[1, 0] = 0
        res, jac = orc.residual_jacobian(Kb, clips, pose, g12['x2d'][b], g12['x3d'][b], g12['istd'][b])
        for j in range(4):
            h = 1e-6
            pp, pm = pose.copy(), pose.copy(); pp[j] += h; pm[j] -= h
            rp, _ = orc.residual_jacobian(Kb, clips, pp, g12['x2d'][b], g12['x3d'][b], g12['istd'][b])
            rm, _ = orc.residual_jacobian(Kb, clips, pm, g12['x2d'][b], g12['x3d'][b], g12['istd'][b])
            fd = (rp - rm) / (2 * h)
            # points whose clamp state flips inside the FD stencil are excluded
            _, jp = orc.residual_jacobian(Kb, clips, pp, g12['x2d'][b], g12['x3d'][b], g12['istd'][b])
            _, jm = orc.residual_jacobian(Kb, clips, pm, g12['x2d'][b], g12['x3d'][b], g12['istd'][b])
            stable = ((jp[:, :, j] == 0) == (jm[:, :, j] == 0))
            d = np.abs(fd - jac[:, :, j])[stable]
            assert d.max() <= 1e-5 * max(1.0, np.abs(jac[:, :, j]).max()), (b, j, d.max())


def test_clamp_derivative_semantics_differ_between_solver_and_covariance(orc, g12):
    """under a z-clamp Ceres keeps d u/d tx = fx/z_min while torch zeroes the whole point."""
    b = 1
    clips = [float(g12['z_min']), *g12['u_range'][b], *g12['v_range'][b]]
    pose = np.array([g12['yaw'][b], *g12['t'][b]])
    _, jac = orc.residual_jacobian(g12['K'][b], clips, pose, g12['x2d'][b], g12['x3d'][b], g12['istd'][b])
    tj, _, _ = orc.torch_jacobian(g12['K'][b], float(g12['z_min']), g12['u_range'][b], g12['v_range'][b],
                                  g12['yaw'][b], g12['t'][b], g12['x2d'][b], g12['x3d'][b], g12['istd'][b], None)
    c, s = np.cos(pose[0]), np.sin(pose[0])
    Z = -s * g12['x3d'][b][:, 0] + c * g12['x3d'][b][:, 2] + pose[3]
    zc = Z < 0.5
    assert zc.sum() > 50
    uin = (np.abs(jac[zc][:, 0, 1]) > 0)
    assert uin.any()
    sel = np.where(zc)[0][uin]
    np.testing.assert_allclose(jac[sel, 0, 1], g12['istd'][b][sel, 0] * g12['K'][b][0, 0] / 0.5, rtol=1e-12)
    assert np.all(jac[sel, 0, 3] == 0) and np.all(tj[zc] == 0)


# ----------------------------------------------------------------------------- decode chain ---
def test_slice_pred_channel_indexing_is_bit_exact(orc, g3):
    noc, logstd, chan = orc.slice_pred(g3['all_pred'], g3['labels'], g3['flip'], num_classes=3)
    assert np.array_equal(noc, g3['noc_pred']) and np.array_equal(logstd, g3['proj_logstd'])
    C = 3
    f, c = g3['flip'].astype(int), g3['labels']
    for k in range(3):
        assert np.array_equal(chan[:, k], f * 5 * C + 3 * c + k)
    for k in range(2):
        assert np.array_equal(chan[:, 3 + k], f * 5 * C + 3 * C + 2 * c + k)
    noc_a, ls_a, _ = orc.slice_pred(g3['all_pred'][:, :10], None, g3['flip'], class_agnostic=True)
    assert np.array_equal(noc_a, g3['noc_agnostic']) and np.array_equal(ls_a, g3['logstd_agnostic'])


def test_decode_chain_matches_reference_coders(orc, g3):
    dims, dvar = orc.dim_decode(g3['dim'], g3['dim_var'], g3['labels'])
    np.testing.assert_allclose(dims, g3['dims'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dvar, g3['dims_var'], rtol=1e-6, atol=1e-9)
    c3d, c3v = orc.noc_decode(g3['noc_pred'], dims, dvar)
    np.testing.assert_allclose(c3d, g3['c3d'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c3v, g3['c3d_var'], rtol=1e-5, atol=1e-9)
    ls = orc.decode_logstd(g3['proj_logstd'], c3v)
    np.testing.assert_allclose(ls, g3['logstd_px'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(orc.decode_logstd(g3['proj_logstd'], None), g3['logstd_px_novar'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(orc.cov_correction(g3['cov_in'], g3['tvec']), g3['cov_corr'], rtol=1e-5)


def test_pose_head_prep_matches_reference(orc, g4):
    x2d, istd, x3d, ur, vr, thr = orc.pose_head_prep(g4['coords_2d'], g4['coords_2d_logstd'], g4['coords_3d'], g4['img_shapes'])
    assert np.array_equal(x2d, g4['pnp_coords_2d']) and np.array_equal(x3d, g4['pnp_coords_3d'])      # pure index permutation
    np.testing.assert_allclose(istd, g4['pnp_istd'], rtol=2e-6)
    assert np.array_equal(ur, g4['u_range']) and np.array_equal(vr, g4['v_range'])
    np.testing.assert_allclose(thr, g4['ransac_thr'], rtol=1e-6)
    # the reference hands the PnP channel-planar *views*: strides (C*P, 1, P)
    assert [tuple(s) for s in g4['pnp_strides']] == [(1568, 1, 784), (1568, 1, 784), (2352, 1, 784)]
    assert x2d.strides == (1568 * 4, 4, 784 * 4)
    np.testing.assert_allclose(orc.cov_calib(g4['pose_cov'], g4['cov_calib_logscale']), g4['pose_cov_calib'], rtol=1e-6)


def test_roi_grid_reproduces_fixture_grid(orc, g4):
    c2d = g4['coords_2d']
    x1 = c2d[:, 0, 0, 0] + 0.5 - 0.5 * (c2d[:, 0, 0, 1] - c2d[:, 0, 0, 0])
    bw = (c2d[:, 0, 0, 1] - c2d[:, 0, 0, 0]) * 28
    y1 = c2d[:, 1, 0, 0] + 0.5 - 0.5 * (c2d[:, 1, 1, 0] - c2d[:, 1, 0, 0])
    bh = (c2d[:, 1, 1, 0] - c2d[:, 1, 0, 0]) * 28
    g = orc.roi_grid(np.stack([x1, y1, x1 + bw, y1 + bh], 1))
    np.testing.assert_allclose(g, c2d, rtol=0, atol=2e-3)


@pytest.mark.parametrize('P', [784, 3136, 100, 9])
def test_istd_mask_numpy_orders(orc, P):
    """R4 is literally numpy: the float32 summation order depends on the strides (sequential for a
    C-contiguous (B,P,2) array, pairwise when the point axis is contiguous)."""
    rng = np.random.default_rng(P)
    a = np.exp(rng.normal(size=(5, P, 2))).astype(np.float32)
    planar = np.ascontiguousarray(a.transpose(0, 2, 1)).transpose(0, 2, 1)
    m_c = orc.istd_inlier_mask(a, 0.6)
    m_p = orc.istd_inlier_mask(planar, 0.6)
    seq = np.zeros((5, 2), np.float32)
    for p in range(P):
        seq += a[:, p, :]
    thr = np.float32(0.6) * (seq / np.float32(P))
    assert np.array_equal(m_c, (a >= thr[:, None, :]).all(2))
    assert m_p.shape == m_c.shape and (m_p != m_c).mean() < 0.01


def test_specified_covariance_hessian_equals_the_reference_hessian_to_rounding(orc, g12):
    """Round 6: the covariance Hessian the kernel's stage 4 specifies (orc_cov_hessian_spec: fixed operations, the workgroup's summation
    tree for its wave count) IS the J^T J of jacobian.py / hessian.py:84-86 — it equals the reference's own fp64 output (fixture G2, masked)
    to rounding for every wave count, with clamped points and partial masks in the set; different wave counts are different trees."""
    B = g12['x2d'].shape[0]
    differ = False
    for b in range(B):
        Hs = [orc.cov_hessian_spec(g12['K'][b], float(g12['z_min']), g12['u_range'][b], g12['v_range'][b], g12['yaw'][b], g12['t'][b],
                                   g12['x3d'][b], g12['istd'][b], g12['inlier'][b], waves=w) for w in (1, 2, 4, 8)]
        for H in Hs:
            assert np.array_equal(H, H.T)
            assert _rel(H, g12['h_f64'][b]) <= 1e-12, b
        differ = differ or any(not np.array_equal(Hs[0], H) for H in Hs[1:])
    assert differ
    sn, cs = orc.spec_sincos(np.linspace(-7.0, 7.0, 2001))
    assert np.abs(sn - np.sin(np.linspace(-7.0, 7.0, 2001))).max() <= 2.3e-16 and np.abs(cs - np.cos(np.linspace(-7.0, 7.0, 2001))).max() <= 2.3e-16

"""GPU tests of K2 (fused NOC-head post-processing) and of the pose-head mirror against the golden
vectors produced by the reference's coders / slice_pred / UncertPropPnPOptimizer (G3, G4)."""
import numpy as np
import pytest
import torch

# the one-launch FAST MODE these tests exercise (decode fused into the PnP kernel runs this repository's K0 initialiser); a head built from
# the reference's config dict runs the reference's flow since round 5 (tests/test_gpu_epnp.py, test_reference_flow_* below)
K0_PNP = dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, forward_exact_hessian=False, initialiser='k0')

from monorun_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def test_k2_against_reference_decode_chain(dev, g3, orc):
    from monorun_amd.pose_head import noc_decode
    rng = np.random.default_rng(3)
    B = g3['all_pred'].shape[0]
    rois = np.stack([rng.uniform(0, 900, B), rng.uniform(0, 200, B)], 1)
    rois = np.concatenate([rois, rois + rng.uniform(20, 200, (B, 2))], 1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    dec = noc_decode(t(g3['all_pred']), t(g3['labels']), t(g3['flip']), t(g3['dim']), t(g3['dim_var']), t(rois))
    torch.cuda.synchronize()
    c3d = dec['coords_3d'].cpu().numpy()
    # integer channel pick + unfused fp32 mul/add chain: bit-exact against the reference's own output
    assert np.array_equal(c3d, g3['c3d'])
    assert np.array_equal(dec['dims'].cpu().numpy(), g3['dims']) and np.array_equal(dec['dims_var'].cpu().numpy(), g3['dims_var'])
    # exp/log: the reference's torch.exp / torch.log are library routines (a few ulp from any other library) ...
    istd_ref = np.exp(-g3['logstd_px']) / np.float32(10)
    np.testing.assert_allclose(dec['coords_2d_istd'].cpu().numpy(), istd_ref, rtol=3e-6)
    # ... the kernel's are SPECIFIED (mr_expf / mr_logf: fixed float32 operation sequences) and restated operation for operation
    # in the oracle, so the decoded istd — which feeds a bit-exact threshold — is bit-identical
    n_noc, n_ls, _ = orc.slice_pred(g3['all_pred'], g3['labels'], g3['flip'])
    d_, dv_ = orc.dim_decode(g3['dim'], g3['dim_var'], g3['labels'])
    ls_spec = orc.decode_logstd(n_ls, orc.noc_decode(n_noc, d_, dv_)[1], exp=orc.spec_expf, log=orc.spec_logf)
    assert np.array_equal(dec['coords_2d_istd'].cpu().numpy(), orc.spec_expf(-ls_spec) / np.float32(10))
    grid = orc.roi_grid(rois)
    assert np.array_equal(dec['coords_2d'].cpu().numpy(), grid)
    thr_ref = np.float32(0.2) * (grid[:, 1, -1, 0] - grid[:, 1, 0, 0])
    assert np.array_equal(dec['ransac_thr'].cpu().numpy(), thr_ref)
    # class-agnostic head (kitti_car.py) and no dim variance
    dec_a = noc_decode(t(g3['all_pred'][:, :10]), t(np.zeros(B, np.int64)), t(g3['flip']), t(g3['dim']), None, t(rois),
                       num_classes=1, class_agnostic=True)
    noc_a = g3['noc_agnostic']
    dims0 = g3['dim'] * np.float32([0.44, 0.14, 0.11]) + np.float32([3.89, 1.53, 1.62])
    part = noc_a * np.float32([0.35, 0.23, 0.34])[:, None, None] + np.float32([-0.1, -0.5, 0.0])[:, None, None]
    assert np.array_equal(dec_a['coords_3d'].cpu().numpy(), part * dims0[:, :, None, None])
    np.testing.assert_allclose(dec_a['coords_2d_istd'].cpu().numpy(), np.exp(-g3['logstd_agnostic']) / np.float32(10), rtol=3e-6)
    assert dec_a['dims_var'] is None
    # bool flip for the whole batch + (B,5) mmdet rois
    rois5 = np.concatenate([np.zeros((B, 1), np.float32), rois], 1)
    dec_f = noc_decode(t(g3['all_pred']), t(g3['labels']), True, t(g3['dim']), t(g3['dim_var']), t(rois5))
    noc_f, _, _ = orc.slice_pred(g3['all_pred'], g3['labels'], True)
    d, dv = orc.dim_decode(g3['dim'], g3['dim_var'], g3['labels'])
    assert np.array_equal(dec_f['coords_3d'].cpu().numpy(), orc.noc_decode(noc_f, d, dv)[0])
    assert np.array_equal(dec_f['coords_2d'].cpu().numpy(), grid)


def test_pose_head_mirror_against_reference_prep(dev, g4, orc):
    """The tensors our head hands to the PnP equal what the reference's head handed to its PnP (G4)."""
    from monorun_amd.pose_head import UncertPropPnPOptimizer
    rec = {}

    class Recorder(torch.nn.Module):
        def forward(self, *a):
            rec['a'] = a
            n = a[0].shape[0]
            return (torch.ones(n, dtype=torch.bool, device=a[0].device), a[0].new_zeros(n, 1), a[0].new_zeros(n, 3),
                    torch.from_numpy(g4['pose_cov']).to(a[0].device), None)
    head = UncertPropPnPOptimizer(
        pnp=dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, forward_exact_hessian=False),
        rotation_coder=dict(type='Vec2DRotationCoder'), allowed_border=200, epnp_ransac_thres_ratio=0.2).to(dev)
    assert list(head.state_dict()) == ['cov_calib_logscale']          # the only hot-path state in a checkpoint
    head.pnp = Recorder()
    with torch.no_grad():
        head.cov_calib_logscale.copy_(torch.from_numpy(g4['cov_calib_logscale']))
        t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
        out = head(t(g4['coords_2d']), t(g4['coords_2d_logstd']), t(g4['coords_3d']), t(g4['cam']), t(g4['img_shapes']))
    a = rec['a']
    assert np.array_equal(a[0].cpu().numpy(), g4['pnp_coords_2d']) and np.array_equal(a[2].cpu().numpy(), g4['pnp_coords_3d'])
    assert [tuple(x.stride()) for x in a[:3]] == [tuple(s) for s in g4['pnp_strides']]
    np.testing.assert_allclose(a[1].cpu().numpy(), g4['pnp_istd'], rtol=3e-6)
    assert np.array_equal(a[4].cpu().numpy(), g4['u_range']) and np.array_equal(a[5].cpu().numpy(), g4['v_range'])
    np.testing.assert_allclose(a[6].cpu().numpy(), g4['ransac_thr'], rtol=1e-6)
    np.testing.assert_allclose(out[4].cpu().numpy(), g4['pose_cov_calib'], rtol=2e-6)


def test_tail_end_to_end_two_launches(dev, orc):
    """all_pred -> K2 -> fused PnP -> calib -> cov_correction, against the oracle fed by the numpy
    restatement of the same chain."""
    from monorun_amd.pose_head import UncertPropPnPOptimizer, pose_from_head
    b = syn.make_batch(B=48, seed=99)
    B = 48
    rng = np.random.default_rng(5)
    labels, flip = b['labels'], rng.uniform(size=B) < 0.5
    C = 3
    # invert the decode chain to synthesise a head output that decodes to the batch
    mu, sd = orc.DIM_MEANS[labels], orc.DIM_STDS[labels]
    dim = ((b['dims'] - mu) / sd).astype(np.float32)
    dim_var = (rng.uniform(0.02, 0.1, (B, 3)) ** 2).astype(np.float32)
    dims, dims_var = orc.dim_decode(dim, dim_var, labels)
    noc = ((b['coords_3d'] / dims[:, :, None, None] - orc.NOC_MEANS[:, None, None]) / orc.NOC_STDS[:, None, None]).astype(np.float32)
    all_pred = rng.normal(0, 1, (B, 2 * C * 5, 28, 28)).astype(np.float32)
    _, _, chan = orc.slice_pred(all_pred, labels, flip)
    ar = np.arange(B)
    for k in range(3):
        all_pred[ar, chan[:, k]] = noc[:, k]
    for k in range(2):
        all_pred[ar, chan[:, 3 + k]] = (b['logstd'][:, k] - np.log(2.0)).astype(np.float32) * 0.5
    # oracle chain (numpy restatement, pinned to the reference by the G3/G4 tests)
    n_noc, n_ls, _ = orc.slice_pred(all_pred, labels, flip)
    c3d, c3v = orc.noc_decode(n_noc, dims, dims_var)
    # exp / log as specified for the decode (bit-identical to the kernel's): the istd map, hence the thresholded mask, is exact
    ls_px = orc.decode_logstd(n_ls, c3v, exp=orc.spec_expf, log=orc.spec_logf)
    c2d = orc.roi_grid(b['rois'])
    x2d, istd, x3d, ur, vr, thr = orc.pose_head_prep(c2d, ls_px, c3d, b['img_shape'], exp=orc.spec_expf)
    ref = orc.u2d_pnp(x2d, istd, x3d, b['K'], ur, vr, 0.5, 0.6, thr, True)
    logscale = np.array([0.3, -0.2, 0.1, 0.5], np.float32)
    ref_calib = orc.cov_correction(orc.cov_calib(ref[3], logscale), ref[2])
    # product
    head = UncertPropPnPOptimizer(pnp=K0_PNP).to(dev)
    with torch.no_grad():
        head.cov_calib_logscale.copy_(torch.from_numpy(logscale))
        t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
        res = pose_from_head(head, t(all_pred), t(labels), t(flip), t(dim), t(dim_var), t(b['rois']), t(b['K']), b['img_shape'])
        res2 = pose_from_head(head, t(all_pred), t(labels), t(flip), t(dim), t(dim_var), t(b['rois']), t(b['K']), b['img_shape'], fused=False)
    torch.cuda.synchronize()
    # ONE fused launch == K2 followed by the PnP kernel, bit for bit (the calibrated covariance comes from the kernel's
    # epilogue in the fused path and from torch ops in the other: same float32 formula, ulp-level agreement)
    for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_pred', 'dimensions_pred', 'dimensions_var', 'inlier_mask'):
        assert torch.equal(res[k], res2[k]), k
    torch.testing.assert_close(res['pose_cov_calib'][res['ret_val']], res2['pose_cov_calib'][res2['ret_val']], rtol=2e-6, atol=0)
    assert np.array_equal(res['ret_val'].cpu().numpy(), ref[0])
    ok = ref[0]
    # north-star bars on the whole head -> pose path: inlier masks bit-exact, rotation / translation within 1e-4
    assert np.array_equal(res['inlier_mask'].cpu().numpy(), ref[5])
    assert np.abs(res['t_vec_pred'].cpu().numpy() - ref[2])[ok].max() <= 1e-4
    dy = np.abs(np.angle(np.exp(1j * (res['yaw_pred'].cpu().numpy() - ref[1]))))[ok]
    assert dy.max() <= 1e-4
    rc = res['pose_cov_calib'].cpu().numpy()
    assert (np.abs(rc - ref_calib)[ok] / np.abs(ref_calib)[ok].max((1, 2), keepdims=True)).max() <= 1e-4


def test_fused_head_to_pose_matches_two_launches_everywhere(dev, g3):
    """mr_pnp_from_head_batched (decode inside the PnP kernel, maps never in HBM) against noc_decode + PnPUncert
    on the golden head output: identical valid / pose / cov / inlier mask / dims, incl. class-agnostic and no-variance."""
    from monorun_amd.pose_head import noc_decode, pnp_from_head, _planar_view
    from monorun_amd.ops import pnp_uncert
    rng = np.random.default_rng(8)
    B = g3['all_pred'].shape[0]
    rois = np.stack([rng.uniform(100, 900, B), rng.uniform(50, 200, B)], 1)
    rois = np.concatenate([rois, rois + rng.uniform(30, 200, (B, 2))], 1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    K = t(syn.KITTI_K[None].astype(np.float32))
    img = np.array([[375.0, 1242.0]], np.float32)
    for agn, dv in ((False, g3['dim_var']), (False, None), (True, g3['dim_var'])):
        ap = g3['all_pred'][:, :10] if agn else g3['all_pred']
        lab = np.zeros(B, np.int64) if agn else g3['labels']
        kw = dict(num_classes=1 if agn else 3, class_agnostic=agn)
        out = pnp_from_head(t(ap), t(lab), t(g3['flip']), t(g3['dim']), t(dv) if dv is not None else None, t(rois), K, img,
                            with_diag=True, **kw)
        dec = noc_decode(t(ap), t(lab), t(g3['flip']), t(g3['dim']), t(dv) if dv is not None else None, t(rois), **kw)
        ur = torch.tensor([[-200.0, 1442.0]], device=dev); vr = torch.tensor([[-200.0, 575.0]], device=dev)
        ref = pnp_uncert(_planar_view(dec['coords_2d']), _planar_view(dec['coords_2d_istd']), _planar_view(dec['coords_3d']), K, ur, vr,
                         0.5, 0.6, dec['ransac_thr'], True, initialiser='k0')
        torch.cuda.synchronize()
        for a, b_ in zip(out[:5], ref):
            assert torch.equal(a, b_)
        assert torch.equal(out[5], dec['dims']) and (dv is None or torch.equal(out[6], dec['dims_var']))


@pytest.mark.gpu
def test_fused_calibration_and_distance_correction(dev, g3):
    """The kernel's optional epilogue = UncertPropPnPOptimizer._calibrate (uncert_prop_pnp_optimizer.py:96-97) followed by
    cov_correction (monorun_roi_head.py:530-534) done with torch ops on the same covariance."""
    from monorun_amd import pose_head as ph
    rng = np.random.default_rng(9)
    B = g3['all_pred'].shape[0]
    rois = np.stack([rng.uniform(100, 900, B), rng.uniform(50, 200, B)], 1)
    rois = np.concatenate([rois, rois + rng.uniform(30, 200, (B, 2))], 1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    K = t(syn.KITTI_K[None].astype(np.float32))
    head = ph.UncertPropPnPOptimizer(pnp=K0_PNP).to(dev)
    with torch.no_grad():
        head.cov_calib_logscale.copy_(torch.tensor([0.3, -0.2, 0.1, 0.45]))
    args = (t(g3['all_pred']), t(g3['labels']), t(g3['flip']), t(g3['dim']), t(g3['dim_var']), t(rois), K, (375, 1242))
    for corr in (True, False):
        with torch.no_grad():
            a = ph.pose_from_head(head, *args, apply_cov_correction=corr, fused=True)
            b = ph.pose_from_head(head, *args, apply_cov_correction=corr, fused=False)
        for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_pred', 'dimensions_pred', 'dimensions_var'):
            assert torch.equal(a[k], b[k]), k
        assert a['ret_val'].dtype == torch.bool and a['pose_cov_calib'].shape == (B, 4, 4)
        ok = a['ret_val']
        torch.testing.assert_close(a['pose_cov_calib'][ok], b['pose_cov_calib'][ok], rtol=2e-6, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_half_precision_head_outputs(dev, g3, dtype):
    """fp16 / bf16 head outputs (autocast pipelines) are read as they are and widened exactly: results are bit-identical to
    feeding the same values as fp32, in K2 and in the fused kernel."""
    from monorun_amd.pose_head import noc_decode, pnp_from_head
    rng = np.random.default_rng(12)
    B = g3['all_pred'].shape[0]
    rois = np.stack([rng.uniform(100, 900, B), rng.uniform(50, 200, B)], 1)
    rois = np.concatenate([rois, rois + rng.uniform(30, 200, (B, 2))], 1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    ap_lo = t(g3['all_pred']).to(dtype)
    ap_32 = ap_lo.to(torch.float32)
    args = (t(g3['labels']), t(g3['flip']), t(g3['dim']), t(g3['dim_var']), t(rois))
    a, b_ = noc_decode(ap_lo, *args), noc_decode(ap_32, *args)
    for k in ('coords_2d', 'coords_2d_istd', 'coords_3d', 'dims', 'dims_var', 'ransac_thr'):
        assert torch.equal(a[k], b_[k]), k
    K = t(syn.KITTI_K[None].astype(np.float32))
    img = np.array([[375.0, 1242.0]], np.float32)
    o1, o2 = pnp_from_head(ap_lo, *args, K, img), pnp_from_head(ap_32, *args, K, img)
    for x, y in zip(o1, o2):
        assert torch.equal(x, y)


def test_prepared_and_graph_launch_for_the_per_image_regime(dev, orc):
    """PoseFromHeadLaunch (arguments built once over static buffers; optionally a HIP-graph replay) gives exactly what
    pose_from_head gives, also after the static inputs were overwritten with the next image's data."""
    from monorun_amd.pose_head import UncertPropPnPOptimizer, pose_from_head, PoseFromHeadLaunch
    head = UncertPropPnPOptimizer(pnp=K0_PNP).to(dev)
    with torch.no_grad():
        head.cov_calib_logscale.copy_(torch.tensor([0.2, -0.1, 0.0, 0.4]))
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    imgs = []
    for seed in (3, 4):
        b = syn.make_batch(B=60, seed=seed)
        all_pred, dim = syn.encode_head_outputs(b, seed=seed)
        imgs.append((t(all_pred), t(b['labels']), t(dim), t(b['rois']), t(b['K'])))
    shape = (syn.IMG_H, syn.IMG_W, 3)                       # mmdet's img_meta['img_shape']
    a0 = imgs[0]
    prepared = PoseFromHeadLaunch(head, a0[0].clone(), a0[1].clone(), False, a0[2].clone(), None, a0[3].clone(), a0[4], shape)
    graph = PoseFromHeadLaunch(head, a0[0].clone(), a0[1].clone(), False, a0[2].clone(), None, a0[3].clone(), a0[4], shape).capture()
    for ap, lab, dim, rois, K in imgs:
        with torch.no_grad():
            ref = pose_from_head(head, ap, lab, False, dim, None, rois, K, shape)
        for L, go in ((prepared, prepared.run), (graph, graph.replay)):
            L.inputs['all_pred'].copy_(ap); L.inputs['labels'].copy_(lab); L.inputs['dim'].copy_(dim); L.inputs['rois'].copy_(rois)
            out = go()
            torch.cuda.synchronize()
            for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_pred', 'pose_cov_calib', 'dimensions_pred', 'inlier_mask'):
                assert torch.equal(out[k], ref[k]), k
        assert int(ref['ret_val'].sum()) >= 55


def test_decode_of_more_than_65535_objects_in_one_call(dev, orc):
    """config 5's 65 536-object batch must decode unfused in ONE call (round 1 launched a 2-D grid and refused B > 65 535):
    70 000 objects on a 4x4 RoI grid, checked against the numpy chain at the ends of the batch."""
    from monorun_amd.pose_head import noc_decode
    rng = np.random.default_rng(12)
    B, C = 70000, 3
    all_pred = torch.from_numpy(rng.normal(0, 1, (B, 2 * C * 5, 4, 4)).astype(np.float32)).to(dev)
    labels = rng.integers(0, C, B); dim = rng.normal(0, 1, (B, 3)).astype(np.float32)
    rois = np.stack([rng.uniform(0, 900, B), rng.uniform(0, 200, B)], 1)
    rois = np.concatenate([rois, rois + rng.uniform(20, 200, (B, 2))], 1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    dec = noc_decode(all_pred, t(labels), False, t(dim), None, t(rois))
    torch.cuda.synchronize()
    for sl in (slice(0, 64), slice(65500, 65600), slice(B - 64, B)):
        ap = all_pred[sl].cpu().numpy()
        n_noc, n_ls, _ = orc.slice_pred(ap, labels[sl], False)
        d, _ = orc.dim_decode(dim[sl], None, labels[sl])
        assert np.array_equal(dec['coords_3d'][sl].cpu().numpy(), orc.noc_decode(n_noc, d, None)[0])
        assert np.array_equal(dec['coords_2d_istd'][sl].cpu().numpy(), orc.spec_expf(-orc.decode_logstd(n_ls, None)) / np.float32(10))
        assert np.array_equal(dec['coords_2d'][sl].cpu().numpy(), orc.roi_grid(rois[sl], 4, 4))


@pytest.mark.gpu
def test_pose_head_honours_module_options_the_one_launch_kernel_lacks(dev, orc):
    """A head built with forward_exact_hessian=True (or coord_istd_normalize=True) must not silently get the one-launch kernel's
    J^T J covariance: pose_from_head takes the module path for it (same poses and masks, pose_cov = inverse of the exact Hessian),
    the prepared launch refuses; use_6dof stays ignored inside the head, as in the reference (pnp_uncert.py:11)."""
    import warnings
    from monorun_amd.pose_head import UncertPropPnPOptimizer, pose_from_head, PoseFromHeadLaunch
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    b = syn.make_batch(B=40, seed=8)
    all_pred, dim = syn.encode_head_outputs(b, seed=8)
    args = (t(all_pred), t(b['labels']), False, t(dim), None, t(b['rois']), t(b['K']), (syn.IMG_H, syn.IMG_W, 3))
    cfg = dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, initialiser='k0')
    plain = UncertPropPnPOptimizer(pnp=dict(cfg, forward_exact_hessian=False)).to(dev)
    exact = UncertPropPnPOptimizer(pnp=dict(cfg, forward_exact_hessian=True)).to(dev)
    with torch.no_grad():
        r0, r1 = pose_from_head(plain, *args), pose_from_head(exact, *args)
        r1b = pose_from_head(exact, *args, fused=False)
    for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'inlier_mask', 'dimensions_pred'):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.equal(r1['pose_cov_pred'], r1b['pose_cov_pred']) and torch.equal(r1['pose_cov_calib'], r1b['pose_cov_calib'])
    ok = r0['ret_val'].cpu().numpy()
    assert ok.sum() >= 36
    d = (r1['pose_cov_pred'] - r0['pose_cov_pred']).abs().amax(dim=(1, 2)) / r0['pose_cov_pred'].abs().amax(dim=(1, 2))
    assert float(d[torch.from_numpy(ok).to(dev)].min()) > 1e-5          # a different matrix for every object ...
    # ... namely the inverse of the exact Hessian at the returned pose
    from monorun_amd.pose_head import noc_decode
    dec = noc_decode(*args[:6])
    x2d = dec['coords_2d'].flatten(2).permute(0, 2, 1).cpu().numpy(); x3d = dec['coords_3d'].flatten(2).permute(0, 2, 1).cpu().numpy()
    istd = dec['coords_2d_istd'].flatten(2).permute(0, 2, 1).cpu().numpy()
    yaw, tv, mask, cov = [r1[k].cpu().numpy() for k in ('yaw_pred', 't_vec_pred', 'inlier_mask', 'pose_cov_pred')]
    for i in np.nonzero(ok)[0][:12]:
        H = orc.exact_hessian(b['K'][0] if np.ndim(b['K']) == 3 else b['K'], 0.5, [-200, syn.IMG_W + 200], [-200, syn.IMG_H + 200],
                              float(yaw[i, 0]), tv[i], x2d[i], x3d[i], istd[i], mask[i])
        good, co = orc.pose_cov_general(H)
        assert good and np.abs(cov[i] - co).max() <= 1e-4 * np.abs(co).max(), i
    with pytest.raises(ValueError, match='forward_exact_hessian'):
        PoseFromHeadLaunch(exact, *args)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        six = UncertPropPnPOptimizer(pnp=dict(cfg, use_6dof=True)).to(dev)
    assert any('use_6dof' in str(x.message) for x in w) and six.pnp.use_6dof is False
    with torch.no_grad():
        r6 = pose_from_head(six, *args)
    assert torch.equal(r6['pose_cov_pred'], r0['pose_cov_pred'])


@pytest.mark.gpu
def test_fused_head_to_pose_on_70000_objects(dev):
    """The one-launch head->pose kernel on 70 144 objects (6.6 GB of raw head output, offsets beyond 2^32 bytes): 256 distinct
    objects tiled 274x — the first tile equals the 256-object call bit for bit, and so does every other tile."""
    from monorun_amd.pose_head import UncertPropPnPOptimizer, pose_from_head
    nd, rep = 256, 274
    b = syn.make_batch(B=nd, seed=21)
    all_pred, dim = syn.encode_head_outputs(b, seed=21)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    head = UncertPropPnPOptimizer(pnp=K0_PNP).to(dev)
    shape = (syn.IMG_H, syn.IMG_W, 3)
    with torch.no_grad():
        small = pose_from_head(head, t(all_pred), t(b['labels']), False, t(dim), None, t(b['rois']), t(b['K']), shape)
        big = pose_from_head(head, t(all_pred).repeat(rep, 1, 1, 1), t(b['labels']).repeat(rep), False, t(dim).repeat(rep, 1), None,
                             t(b['rois']).repeat(rep, 1), t(b['K']), shape)
    torch.cuda.synchronize()
    assert big['ret_val'].shape[0] == nd * rep and int(small['ret_val'].sum()) >= 250
    for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_pred', 'pose_cov_calib', 'dimensions_pred', 'inlier_mask'):
        tiles = big[k].view(rep, nd, *big[k].shape[1:])
        assert torch.equal(tiles[0], small[k]), k
        assert bool((tiles == tiles[:1]).all()), k


@pytest.mark.gpu
def test_fused_path_on_56x56_tiles(dev):
    """The config-5 RoI resolution through the pose head: 100 KB tiles, for which the library picks 8 waves per object
    (at most two workgroups fit a CU) — one launch equals decode + PnP launch bit for bit."""
    from monorun_amd.pose_head import UncertPropPnPOptimizer, pose_from_head
    b = syn.make_batch(B=64, hw=56, seed=5)
    all_pred, dim = syn.encode_head_outputs(b, seed=5)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    head = UncertPropPnPOptimizer(pnp=K0_PNP).to(dev)
    args = (head, t(all_pred), t(b['labels']), False, t(dim), None, t(b['rois']), t(b['K']), (syn.IMG_H, syn.IMG_W, 3))
    with torch.no_grad():
        r1, r2 = pose_from_head(*args), pose_from_head(*args, fused=False)
    torch.cuda.synchronize()
    assert int(r1['ret_val'].sum()) >= 60
    for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_pred', 'inlier_mask', 'dimensions_pred'):
        assert torch.equal(r1[k], r2[k]), k


def test_prepared_launches_keep_a_converted_coord_map_alive(dev):
    """A coord_2d map that is not already fp32 / contiguous / on the device is CONVERTED for the kernel; a prepared launch reads
    that copy on every run() / replay(), so it must own it (the copy was released when __init__ returned).  An fp16
    and a non-contiguous fp64 map must give what the eager call gives, also after other allocations have churned the pool."""
    from monorun_amd.pose_head import UncertPropPnPOptimizer, pose_from_head, PoseFromHeadLaunch, NocDecodeLaunch, noc_decode
    head = UncertPropPnPOptimizer(pnp=K0_PNP).to(dev)
    b = syn.make_batch(B=40, seed=11)
    all_pred, dim = syn.encode_head_outputs(b, seed=11)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    ap, lab, dm, rois, K = t(all_pred), t(b['labels']), t(dim), t(b['rois']), t(b['K'])
    H, W = 384, 1248
    ys, xs = np.mgrid[0:H, 0:W]
    ident = np.stack([xs, ys]).astype(np.float32)
    maps = {'fp16': t(ident).to(torch.float16),                                   # exact below 2048
            'fp64 non-contiguous': t(np.stack([xs, ys], -1).astype(np.float64)).permute(2, 0, 1)}
    for name, m in maps.items():
        assert name != 'fp64 non-contiguous' or not m.is_contiguous()
        with torch.no_grad():
            ref = pose_from_head(head, ap, lab, False, dm, None, rois, K, (syn.IMG_H, syn.IMG_W), coord_2d=m)
        dref = noc_decode(ap, lab, False, dm, None, rois, coord_2d=m)
        prepared = PoseFromHeadLaunch(head, ap, lab, False, dm, None, rois, K, (syn.IMG_H, syn.IMG_W), coord_2d=m)
        k2 = NocDecodeLaunch(ap, lab, False, dm, None, rois, coord_2d=m)
        junk = [torch.full((2, H, W), float('nan'), device=dev) for _ in range(8)]     # would land on a released copy's memory
        torch.cuda.synchronize()
        for _ in range(2):
            out = prepared.run()
            dec = k2.run()
            torch.cuda.synchronize()
            for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_calib', 'inlier_mask'):
                assert torch.equal(out[k], ref[k]), (name, k)
            for k in ('coords_2d', 'coords_2d_istd', 'coords_3d', 'ransac_thr'):
                assert torch.equal(dec[k], dref[k]), (name, k)
        del junk
        assert int(ref['ret_val'].sum()) >= 36


def test_pipelined_launches_equal_isolated_launches(dev):
    """PnPPipeline: prepared launches issued round-robin on internal streams (several batches in flight) write exactly what the
    same launches write one at a time; the returned events order a consumer stream behind each result."""
    from monorun_amd import PnPLaunch, PnPPipeline
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    batches = []
    for seed in (21, 22, 23, 24, 25, 26):
        b = syn.make_batch(B=256, seed=seed)
        x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=False)
        batches.append((t(x2d), t(istd), t(x3d), t(K), t(ur), t(vr), t(thr)))
    mk = lambda a: PnPLaunch(*a[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=a[6], inlier_opt_only=True)
    ref = [mk(a) for a in batches]
    for r in ref:
        r.run()
    torch.cuda.synchronize()
    for depth in (1, 3, 4):
        pipe = PnPPipeline(dev, depth=depth)
        ls = [mk(a) for a in batches]
        for rep in range(3):                                   # re-submission of the same launch objects: pinned slots
            evs = [pipe.submit(l, slot=i) for i, l in enumerate(ls)]
        consumer = torch.cuda.Stream(device=dev)
        sums = []
        with torch.cuda.stream(consumer):
            for l, ev in zip(ls, evs):
                consumer.wait_event(ev)
                sums.append(l.pose.double().sum())
        consumer.synchronize()
        pipe.drain()
        for l, r, sm in zip(ls, ref, sums):
            assert torch.equal(l.pose, r.pose) and torch.equal(l.cov, r.cov) and torch.equal(l.valid, r.valid) and torch.equal(l.mask, r.mask)
            assert float(sm) == float(r.pose.double().sum())


@pytest.mark.gpu
def test_fused_path_with_one_and_two_waves_per_object(dev, g3):
    """The one- and two-wave instantiations keep a packed tile in LDS (12-byte B records, one index list, a bit mask: six objects per CU
    at P = 784): the fused head -> pose launch through them gives the inlier masks, validity and dims of the four-wave launch bit for bit
    and its poses / covariances to summation-order noise."""
    from monorun_amd.pose_head import pnp_from_head
    rng = np.random.default_rng(9)
    B = g3['all_pred'].shape[0]
    rois = np.stack([rng.uniform(100, 900, B), rng.uniform(50, 200, B)], 1)
    rois = np.concatenate([rois, rois + rng.uniform(30, 200, (B, 2))], 1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    K = t(syn.KITTI_K[None].astype(np.float32))
    img = np.array([[375.0, 1242.0]], np.float32)
    args = (t(g3['all_pred']), t(g3['labels']), t(g3['flip']), t(g3['dim']), t(g3['dim_var']), t(rois), K, img)
    ref = pnp_from_head(*args, with_diag=True, flags=4 << 8)
    for wpo in (1, 2):
        out = pnp_from_head(*args, with_diag=True, flags=wpo << 8)
        torch.cuda.synchronize()
        assert torch.equal(out[0], ref[0]) and torch.equal(out[4], ref[4]) and torch.equal(out[5], ref[5]), wpo       # valid, inlier mask, dims
        ok = ref[0]
        assert torch.allclose(out[1][ok], ref[1][ok], rtol=0, atol=1e-5) and torch.allclose(out[2][ok], ref[2][ok], rtol=1e-6, atol=1e-5), wpo
        assert torch.allclose(out[3][ok], ref[3][ok], rtol=1e-4, atol=1e-9), wpo


@pytest.mark.gpu
def test_k2_special_inputs_take_the_exact_path(dev, g3, orc):
    """The special cases of the specified exp / log sequences in the vector decode kernel (packed two-pixel forms with selects): log-std
    values beyond the exp range, infinities, NaNs, non-positive variances planted in a few pixels of otherwise ordinary objects decode
    exactly as the specification (oracle) says."""
    from monorun_amd.pose_head import noc_decode
    rng = np.random.default_rng(11)
    pred = np.array(g3['all_pred'], np.float32, copy=True)
    B, C, H, W = pred.shape
    plant = np.float32([45.0, 60.0, 100.0, 200.0, -45.0, -52.0, -60.0, -200.0, 88.0, -88.0, np.inf, -np.inf, np.nan, 3e38, -3e38, 1e-40])
    for b in range(0, B, 2):                                  # every second object: a handful of pixels in every log-std channel
        for c in range(C):
            for v in plant[rng.integers(0, len(plant), 3)]:
                pred[b, c, rng.integers(0, H), rng.integers(0, W)] = v
    rois = np.stack([rng.uniform(0, 900, B), rng.uniform(0, 200, B)], 1)
    rois = np.concatenate([rois, rois + rng.uniform(20, 200, (B, 2))], 1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    dec = noc_decode(t(pred), t(g3['labels']), t(g3['flip']), t(g3['dim']), t(g3['dim_var']), t(rois))
    torch.cuda.synchronize()
    with np.errstate(all='ignore'):
        n_noc, n_ls, _ = orc.slice_pred(pred, g3['labels'], g3['flip'])
        d_, dv_ = orc.dim_decode(g3['dim'], g3['dim_var'], g3['labels'])
        c3d_ref, var_ref = orc.noc_decode(n_noc, d_, dv_)
        ls_spec = orc.decode_logstd(n_ls, var_ref, exp=orc.spec_expf, log=orc.spec_logf)
        istd_ref = orc.spec_expf(-ls_spec) / np.float32(10)
    got = dec['coords_2d_istd'].cpu().numpy()
    same = (got.view(np.uint32) == istd_ref.view(np.uint32)) | (np.isnan(got) & np.isnan(istd_ref))
    assert same.all(), f'{(~same).sum()} decoded istd values differ from the specification on special inputs'
    assert (~np.isfinite(got)).sum() > 0 and (got == 0).sum() > 0, 'the planted values did not reach the special cases'
    c3d = dec['coords_3d'].cpu().numpy()
    assert ((c3d.view(np.uint32) == c3d_ref.view(np.uint32)) | (np.isnan(c3d) & np.isnan(c3d_ref))).all()


@pytest.mark.gpu
def test_head_built_with_the_reference_initialiser_is_honoured_everywhere(dev, orc):
    """a head whose pnp dict says initialiser='epnp' (the reference's flow: uncert_prop_pnp_optimizer.py:86-95 ->
    pnp_uncert_cpu.py:33-68) gets that flow from every entry point — pose_from_head whatever `fused` says (the one-launch kernel only
    knows K0, so both forms take K2 + module path), the prepared PoseFromHeadLaunch (K2 -> EPnP / RANSAC -> LM, also as a graph) and
    u2d_pnp_cpu — and the result is the CPU restatement's: K2's decode chain -> u2d_pnp_epnp, masks bit-exact, pose within 1e-4."""
    from monorun_amd.pose_head import UncertPropPnPOptimizer, pose_from_head, PoseFromHeadLaunch
    from monorun_amd.ops import u2d_pnp_cpu
    b = syn.make_batch(B=96, seed=41)
    all_pred, dim = syn.encode_head_outputs(b, seed=41)
    n_noc, n_ls, _ = orc.slice_pred(all_pred, b['labels'], False)
    dims, _ = orc.dim_decode(dim, None, b['labels'])
    c3d, _ = orc.noc_decode(n_noc, dims, None)
    x2d, istd, x3d, ur, vr, thr = orc.pose_head_prep(orc.roi_grid(b['rois']), orc.decode_logstd(n_ls, None), c3d, b['img_shape'], exp=orc.spec_expf)
    ref = orc.u2d_pnp_epnp(x2d, istd, x3d, b['K'], ur, vr, 0.5, 0.6, thr, True)
    ref_k0 = orc.u2d_pnp(x2d, istd, x3d, b['K'], ur, vr, 0.5, 0.6, thr, True)
    assert not np.array_equal(ref[5], ref_k0[5])                       # the two initialisers DO differ on this batch: the test can tell them apart
    t = lambda a, dt=None: torch.from_numpy(np.asarray(a)).to(dev) if dt is None else torch.from_numpy(np.asarray(a)).to(device=dev, dtype=dt)
    head = UncertPropPnPOptimizer(pnp=dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, forward_exact_hessian=False)).to(dev)      # the reference's own config dict (configs/kitti_car.py:118-126)
    assert head.pnp.initialiser == 'epnp'
    args = (t(all_pred), t(b['labels']), False, t(dim), None, t(b['rois']), t(b['K']), (syn.IMG_H, syn.IMG_W, 3))

    def check(res, what):
        ok = ref[0]
        assert np.array_equal(res['ret_val'].cpu().numpy(), ref[0]) and np.array_equal(res['inlier_mask'].cpu().numpy(), ref[5]), what
        dyaw = np.abs(np.angle(np.exp(1j * (res['yaw_pred'].cpu().numpy() - ref[1]))))
        assert ok.sum() >= 90 and dyaw[ok].max() <= 1e-4 and np.abs(res['t_vec_pred'].cpu().numpy() - ref[2])[ok].max() <= 1e-4, what
    with torch.no_grad():
        r_f = pose_from_head(head, *args, fused=True)
        r_u = pose_from_head(head, *args, fused=False)
    torch.cuda.synchronize()
    check(r_f, 'pose_from_head(fused=True)'); check(r_u, 'pose_from_head(fused=False)')
    for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_pred', 'pose_cov_calib', 'inlier_mask', 'dimensions_pred'):
        assert torch.equal(r_f[k], r_u[k]), k
    # the prepared launch honours it too (it used to run K0 silently), eagerly and as a captured graph
    pl = PoseFromHeadLaunch(head, *args)
    out = pl.run(); torch.cuda.synchronize()
    check(out, 'PoseFromHeadLaunch.run')
    for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_pred', 'inlier_mask', 'dimensions_pred'):
        assert torch.equal(out[k], r_f[k]), k
    assert torch.allclose(out['pose_cov_calib'][out['ret_val']], r_f['pose_cov_calib'][out['ret_val']], rtol=2e-6, atol=0)      # expf in the kernel's epilogue, torch.exp in the module path
    keep = {k: out[k].clone() for k in ('pose', 'pose_cov_calib', 'inlier_mask_u8')}
    out['pose'].zero_(); out['pose_cov_calib'].zero_(); out['inlier_mask_u8'].zero_()
    pl.replay(); torch.cuda.synchronize()
    assert all(torch.equal(out[k], keep[k]) for k in keep)
    # ... as one launch set with other images' proposals (PoseFromHeadGroupLaunch: K2 per member, the initialiser's launches and the re-fit / LM
    # launch over the objects of all members, calibration per member): every member's outputs equal its own launch's, bit for bit
    from monorun_amd.pose_head import PoseFromHeadGroupLaunch
    b2 = syn.make_batch(B=96, seed=42)
    ap2, dim2 = syn.encode_head_outputs(b2, seed=42)
    args2 = (t(ap2), t(b2['labels']), False, t(dim2), None, t(b2['rois']), t(b2['K']), (syn.IMG_H, syn.IMG_W, 3))
    solo2 = PoseFromHeadLaunch(head, *args2); solo2.run(); torch.cuda.synchronize()
    members = [PoseFromHeadLaunch(head, *args), PoseFromHeadLaunch(head, *args2), PoseFromHeadLaunch(head, *args)]
    outs = PoseFromHeadGroupLaunch(members).run(); torch.cuda.synchronize()
    for o, r in zip(outs, (out, solo2.out, out)):
        for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_pred', 'pose_cov_calib', 'inlier_mask', 'dimensions_pred'):
            assert torch.equal(o[k], r[k]), k
    with pytest.raises(ValueError):
        PoseFromHeadGroupLaunch([PoseFromHeadLaunch(UncertPropPnPOptimizer(pnp=dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, initialiser='k0')).to(dev), *args)])
    # ... and the numpy-level driver the reference's flow is written in
    rr = u2d_pnp_cpu(x2d, istd, x3d, b['K'], ur, vr, 0.5, 0.6, thr, True)
    assert np.array_equal(rr[0], ref[0]) and np.array_equal(rr[5], ref[5])
    assert np.abs(rr[2] - ref[2])[ref[0]].max() <= 1e-4 and np.abs(np.angle(np.exp(1j * (rr[1] - ref[1]))))[ref[0]].max() <= 1e-4
    with pytest.raises(ValueError):
        u2d_pnp_cpu(x2d, istd, x3d, b['K'], ur, vr, 0.5, 0.6, thr, True, initialiser='opencv')


@pytest.mark.gpu
@pytest.mark.parametrize('h,w', [(6, 6), (10, 6), (4, 7), (12, 5), (2, 2), (28, 28), (14, 30)])
def test_k2_vector_kernel_on_maps_whose_rows_do_not_hold_whole_quads(dev, orc, h, w):
    """h * w % 4 == 0 takes the vector kernel (four pixels per lane; the last wave of an object in pairs) also when w % 4 != 0: quads and
    pairs then straddle rows, and the pixel grid (row = p / w by multiplication) must still be the oracle's."""
    from monorun_amd.pose_head import noc_decode
    rng = np.random.default_rng(100 * h + w)
    B, C = 37, 3
    pred = rng.normal(0, 1, (B, 2 * C * 5, h, w)).astype(np.float32)
    labels = rng.integers(0, C, B); flip = rng.integers(0, 2, B).astype(bool)
    dim = rng.normal(0, 1, (B, 3)).astype(np.float32); dim_var = (0.01 * rng.random((B, 3)) + 1e-4).astype(np.float32)
    rois = np.stack([rng.uniform(0, 900, B), rng.uniform(0, 200, B)], 1)
    rois = np.concatenate([rois, rois + rng.uniform(20, 200, (B, 2))], 1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    dec = noc_decode(t(pred), t(labels), t(flip), t(dim), t(dim_var), t(rois))
    torch.cuda.synchronize()
    assert np.array_equal(dec['coords_2d'].cpu().numpy(), orc.roi_grid(rois, h, w))
    n_noc, n_ls, _ = orc.slice_pred(pred, labels, flip)
    d_, dv_ = orc.dim_decode(dim, dim_var, labels)
    c3d_ref, var_ref = orc.noc_decode(n_noc, d_, dv_)
    assert np.array_equal(dec['coords_3d'].cpu().numpy(), c3d_ref)
    ls_spec = orc.decode_logstd(n_ls, var_ref, exp=orc.spec_expf, log=orc.spec_logf)
    assert np.array_equal(dec['coords_2d_istd'].cpu().numpy(), orc.spec_expf(-ls_spec) / np.float32(10))

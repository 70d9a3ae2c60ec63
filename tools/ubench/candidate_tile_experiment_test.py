"""The candidate-tile instantiation of the fused kernel (two waves per object; only the istd candidates' records in LDS) against the
full-tile one: every output bit for bit, on the shapes it takes, the shapes it hands to the full-tile launch behind it, and mixtures."""
import numpy as np
import pytest
import torch

from monorun_amd import synthetic as syn
from monorun_amd import _lib

pytestmark = pytest.mark.gpu
MR_FULL_TILE = 0x80
W2 = 2 << 8


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _dv(a, dev):
    t = torch.from_numpy(np.asarray(a))
    d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev)
    d.copy_(t)
    return d


def _run(dev, args, flags, init=None, inlier_opt_only=True, thres=0.6, use_thr=True):
    from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
    x2d, istd, x3d, K, ur, vr, thr = args
    out = pnp_uncert_device(_dv(x2d, dev), _dv(istd, dev), _dv(x3d, dev), _dv(K, dev), _dv(ur, dev), _dv(vr, dev), z_min=0.5,
                            epnp_istd_thres=thres, epnp_ransac_thres=_dv(thr, dev) if use_thr else None, inlier_opt_only=inlier_opt_only,
                            init_pose=_dv(init, dev) if init is not None else None, flags=flags, with_diag=True)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in out]


def _same(a, b, tag):
    names = ('valid', 'pose', 'cov', 'tr_radius', 'inlier_mask', 'diag')
    for x, y, nm in zip(a, b, names):
        assert x.dtype == y.dtype and x.shape == y.shape
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f'{tag}: {nm} differs in {(x != y).sum()} elements'
    assert set(np.unique(a[0]).tolist()) <= {0, 1}, f'{tag}: a redo mark survived in valid'


def test_header_constant():
    assert _lib.MR_FULL_TILE == MR_FULL_TILE


@pytest.mark.parametrize('planar', [True, False])
def test_config2_candidate_tile_equals_full_tile(dev, planar):
    args = syn.pnp_boundary(syn.make_batch(B=1024, seed=1234), planar=planar)
    _same(_run(dev, args, W2), _run(dev, args, W2 | MR_FULL_TILE), f'planar={planar}')


def test_objects_the_tile_cannot_hold_are_redone(dev):
    """Every third object: all weights equal (every point a candidate: more than the tile holds); every fifth: one huge weight (fewer
    than five candidates: the reference's 'too few -> every point' rule).  Both go to the full-tile launch behind."""
    x2d, istd, x3d, K, ur, vr, thr = [np.array(a) for a in syn.pnp_boundary(syn.make_batch(B=300, seed=77), planar=True)]
    istd = np.ascontiguousarray(istd.transpose(0, 2, 1)).transpose(0, 2, 1)        # writable planar copy
    istd[::3] = 0.05
    istd[::5] = 1e-3
    istd[::5, 7, :] = 50.0
    args = (x2d, istd, x3d, K, ur, vr, thr)
    a, b = _run(dev, args, W2), _run(dev, args, W2 | MR_FULL_TILE)
    _same(a, b, 'mixed')
    assert a[4][0].sum() > 640, 'a consensus set larger than any candidate tile: this object was solved by the full-tile launch'
    assert 0 < a[4][1].sum() < 624


@pytest.mark.parametrize('hw', [(28, 28), (30, 29), (25, 25), (23, 31)])
def test_tile_shapes(dev, hw):
    """P = 784 (config 2), 870 (close to the 896 points the registers hold), 625 and 713 (odd point counts, partial last chunk)."""
    h, w = hw
    rng = np.random.default_rng(h * 100 + w)
    b = syn.make_batch(B=160, hw=max(h, w), seed=5 + h)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    P = h * w
    sel = np.sort(rng.permutation(x2d.shape[1])[:P])
    cut = lambda t: np.ascontiguousarray(np.asarray(t)[:, sel].transpose(0, 2, 1)).transpose(0, 2, 1)
    args = (cut(x2d), cut(istd), cut(x3d), K, ur, vr, thr)
    _same(_run(dev, args, W2), _run(dev, args, W2 | MR_FULL_TILE), f'P={P}')


def test_other_modes_keep_working_with_two_waves(dev):
    """Given initial pose, no RANSAC threshold, all points in the LM (inlier_opt_only=False: the launcher keeps the full tile),
    a strict istd threshold (few candidates) and a lax one (nearly all)."""
    args = syn.pnp_boundary(syn.make_batch(B=256, seed=99), planar=True)
    ref = _run(dev, args, W2 | MR_FULL_TILE)
    init = np.concatenate([ref[1][:, :1], ref[1][:, 1:]], 1).astype(np.float64) + 0.01
    _same(_run(dev, args, W2, init=init), _run(dev, args, W2 | MR_FULL_TILE, init=init), 'init_pose')
    _same(_run(dev, args, W2, use_thr=False, init=init), _run(dev, args, W2 | MR_FULL_TILE, use_thr=False, init=init), 'no ransac')
    _same(_run(dev, args, W2, inlier_opt_only=False), _run(dev, args, W2 | MR_FULL_TILE, inlier_opt_only=False), 'all points in the LM')
    for thres in (1.6, 0.05, 0.9):
        _same(_run(dev, args, W2, thres=thres), _run(dev, args, W2 | MR_FULL_TILE, thres=thres), f'istd_thres={thres}')


def test_large_batch_picks_it_by_default_and_matches_the_oracle(dev, orc):
    """B = 4096 -> the library's own rule picks two waves per object (and with it the candidate tile)."""
    b = syn.make_batch(B=4096, seed=4242)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
    got = _run(dev, (x2d, istd, x3d, K, ur, vr, thr), 0)
    _same(got, _run(dev, (x2d, istd, x3d, K, ur, vr, thr), MR_FULL_TILE), 'B=4096')
    sl = slice(0, 4096, 16)
    ref = orc.u2d_pnp(x2d[sl], istd[sl], x3d[sl], K[sl] if K.shape[0] > 1 else K, ur, vr, 0.5, 0.6, thr[sl], True, return_diag=True, num_threads=0)
    assert np.array_equal(got[4][sl].astype(bool), ref[5]) and np.array_equal(got[0][sl].astype(bool), ref[0])
    assert np.abs(got[1][sl][:, 1:] - ref[2]).max() <= 1e-4

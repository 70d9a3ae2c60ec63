"""`forward_exact_hessian=True` (the reference's hessian.py:5-64 + pnp_uncert.py:63-85).

G8 (tests/golden/g8_exact_hessian.npz, generator make_golden_hessian.py) holds the matrices the REFERENCE's own exact_hessian
returned for seeded inputs: ordinary objects, z-clipped points, u-only and v-only clipped rows, partial / full / six-point
inlier masks, a skewed camera, a camera whose third row is not (0,0,1) (there the reference's h is not symmetric: the
analytic expressions are differentiated as they stand), tight clip ranges, and an object whose h is exactly singular.
CPU: the oracle's closed form reproduces them to fp64 rounding.  GPU: the HIP kernel against G8 and against the oracle."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def g8():
    return dict(np.load(os.path.join(GOLDEN, 'g8_exact_hessian.npz')))


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _orc_h(orc, g, b, masked=True, f32=False):
    c = (lambda a: a.astype(np.float32).astype(np.float64)) if f32 else (lambda a: a)
    return orc.exact_hessian(c(g['K'][b]), float(g['z_min']), c(g['u_range'][b]), c(g['v_range'][b]), float(c(g['yaw'][b])[0]), c(g['t'][b]),
                             c(g['x2d'][b]), c(g['x3d'][b]), c(g['istd'][b]), g['mask'][b] if masked else None)


def test_oracle_reproduces_the_reference_exact_hessian(orc, g8):
    for b in range(10):
        assert _rel(_orc_h(orc, g8, b), g8['h_exact_f64'][b]) <= 1e-13, b
        assert _rel(_orc_h(orc, g8, b, masked=False), g8['h_exact_nomask_f64'][b]) <= 1e-13, b
        # the reference's own fp32 run agrees with its fp64 run only to fp32 accuracy (conditioning of the sums)
        assert _rel(g8['h_exact_f32'][b], g8['h_exact_f64'][b]) <= 2e-4, b


def test_fixture_exercises_what_it_claims(g8):
    he, ha = g8['h_exact_f64'], g8['h_approx_f64']
    rel2 = np.abs(he - ha).max(axis=(1, 2)) / np.abs(ha).max(axis=(1, 2))
    assert rel2.min() > 1e-3 and rel2.max() > 1.0                      # the second-order term is never negligible, sometimes dominant
    asym = np.abs(he - he.transpose(0, 2, 1)).max(axis=(1, 2)) / np.abs(he).max(axis=(1, 2))
    assert asym[7] > 1e-6 and np.delete(asym, 7).max() < 1e-12          # only the camera with a general third row gives an asymmetric h
    assert np.linalg.matrix_rank(he[9]) < 4                            # object 9: every u row clipped -> tx unobservable
    assert g8['mask'][4].all() and g8['mask'][5].sum() <= 6 and g8['K'][6, 0, 1] != 0


def test_general_inverse_matches_torch_inverse(orc, g8):
    for b in range(9):
        ok, cov = orc.pose_cov_general(g8['h_exact_f64'][b])
        assert ok and _rel(cov, g8['cov_exact_f64'][b]) <= 1e-9, b
    ok, cov = orc.pose_cov_general(g8['h_exact_f64'][9])               # exactly singular -> identity + invalid
    assert not ok and np.array_equal(cov, np.eye(4))
    ok, cov = orc.pose_cov_general(np.diag([1.0, -2.0, 4.0, 0.5]))     # indefinite but regular: inverted like torch.inverse does
    assert ok and np.allclose(cov, np.diag([1.0, -0.5, 0.25, 2.0]))


def test_exact_equals_gauss_newton_at_zero_residual(orc):
    """With x2d = projection at the evaluation pose every e_r is 0 and h is J^T J (R7)."""
    rng = np.random.default_rng(3)
    K = np.array([[707.0912, 0, 601.8873], [0, 707.0912, 183.1104], [0, 0, 1.0]])
    X = rng.uniform(-1, 1, (50, 3)) * [2, 0.8, 0.9]
    yaw, t = 0.4, np.array([1.0, 1.5, 12.0])
    R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
    uvz = (X @ R.T + t) @ K.T
    x2d = uvz[:, :2] / uvz[:, 2:]
    w = rng.uniform(0.05, 0.3, (50, 2))
    ur, vr = np.array([-200, 1442.]), np.array([-200, 575.])
    H = orc.exact_hessian(K, 0.5, ur, vr, yaw, t, x2d, X, w)
    _, _, Ha = orc.torch_jacobian(K, 0.5, ur, vr, yaw, t, x2d, X, w)
    assert _rel(H, Ha) <= 1e-10


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_hip_exact_hessian_against_the_reference_fixture(dev, orc, g8, dtype):
    from monorun_amd.ops.least_squares.pnp_uncert import exact_hessian_device
    T = lambda a, dt=dtype: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    valid_in = torch.ones(10, dtype=torch.uint8, device=dev)
    valid, cov, hess = exact_hessian_device(T(g8['x2d']), T(g8['istd']), T(g8['x3d']), T(g8['K']), T(g8['u_range']), T(g8['v_range']),
                                            T(np.concatenate([g8['yaw'], g8['t']], 1), torch.float32), T(g8['mask'], torch.uint8), valid_in, z_min=0.5,
                                            with_hessian=True)
    valid, cov, hess = valid.cpu().numpy(), cov.cpu().numpy().astype(np.float64), hess.cpu().numpy().astype(np.float64)
    f32 = dtype == torch.float32
    for b in range(10):
        # the kernel's arithmetic is fp64 on the values it is GIVEN (inputs in `dtype`, pose / camera / ranges in fp32): against the oracle on
        # the same rounded values it agrees to the float32 rounding of the output ...
        Ho = orc.exact_hessian(g8['K'][b].astype(np.float32), 0.5, g8['u_range'][b].astype(np.float32), g8['v_range'][b].astype(np.float32),
                               float(np.float32(g8['yaw'][b, 0])), g8['t'][b].astype(np.float32),
                               g8['x2d'][b].astype(np.float32) if f32 else g8['x2d'][b], g8['x3d'][b].astype(np.float32) if f32 else g8['x3d'][b],
                               g8['istd'][b].astype(np.float32) if f32 else g8['istd'][b], g8['mask'][b])
        assert _rel(hess[b], Ho) <= 3e-7, (b, _rel(hess[b], Ho))
        # ... and against the reference's fp64 matrices to what rounding the inputs to float32 costs (the reference's own fp32 run: 2e-4)
        assert _rel(hess[b], g8['h_exact_f64'][b]) <= 2e-4, (b, _rel(hess[b], g8['h_exact_f64'][b]))
        if b < 9:
            ok, co = orc.pose_cov_general(Ho)
            assert valid[b] == 1 and ok
            assert _rel(cov[b], co) <= 1e-5 * max(1.0, np.linalg.cond(Ho) * 1e-7), (b, _rel(cov[b], co))
    assert valid[9] == 0 and np.array_equal(cov[9], np.eye(4))        # singular h -> identity, flag cleared
    # an object that enters invalid stays invalid with h = 0, cov = I; a missing mask means all points
    vin = torch.tensor([1, 0, 1], dtype=torch.uint8, device=dev)
    v2, c2, h2 = exact_hessian_device(T(g8['x2d'][:3]), T(g8['istd'][:3]), T(g8['x3d'][:3]), T(g8['K'][:1]), T(g8['u_range'][:1]), T(g8['v_range'][:1]),
                                      T(np.concatenate([g8['yaw'], g8['t']], 1)[:3], torch.float32), None, vin, z_min=0.5, with_hessian=True)
    assert v2.cpu().tolist() == [1, 0, 1] and np.array_equal(c2[1].cpu().numpy(), np.eye(4)) and float(h2[1].abs().max()) == 0.0
    assert _rel(h2[0].cpu().numpy().astype(np.float64), g8['h_exact_nomask_f64'][0]) <= 2e-4


@pytest.mark.gpu
def test_forward_exact_hessian_flag_end_to_end(dev, orc, batch64):
    """pnp_uncert(..., forward_exact_hessian=True): same poses, masks and validity as the default call; pose_cov = inverse of the exact
    Hessian at the returned pose (oracle closed form on the same float32 values); strided head layout and fp16 storage included."""
    from monorun_amd import synthetic as syn
    from monorun_amd.ops import pnp_uncert, PnPUncert
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=True)

    def dv(a):
        t = torch.from_numpy(np.asarray(a))
        d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev)
        d.copy_(t)
        return d
    args = [dv(a) for a in (x2d, istd, x3d, K, ur, vr)]
    base = pnp_uncert(*args, z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=dv(thr), inlier_opt_only=True)
    ex = PnPUncert(forward_exact_hessian=True)(*args, dv(thr))
    for i in (0, 1, 2, 4):
        assert torch.equal(base[i], ex[i])
    ret, yaw, t, cov, mask = [v.cpu().numpy() for v in ex]
    assert ret.sum() >= 60 and not np.allclose(cov, base[3].cpu().numpy())
    worst = 0.0
    for b in np.nonzero(ret)[0]:
        H = orc.exact_hessian(K[0], 0.5, ur[0], vr[0], float(yaw[b, 0]), t[b], np.asarray(x2d[b]), np.asarray(x3d[b]), np.asarray(istd[b]), mask[b])
        ok, co = orc.pose_cov_general(H)
        assert ok
        worst = max(worst, _rel(cov[b].astype(np.float64), co))
    assert worst <= 1e-4, worst


@pytest.mark.gpu
def test_exact_hessian_edge_inputs(dev, orc, batch64):
    """Empty batch, fp16 storage, non-planar (B,P,C) layout and a C-ABI argument check."""
    from monorun_amd import synthetic as syn, _lib
    from monorun_amd.ops import pnp_uncert
    from monorun_amd.ops.least_squares.pnp_uncert import exact_hessian_device
    e = pnp_uncert(torch.zeros(0, 784, 2, device=dev), torch.zeros(0, 784, 2, device=dev), torch.zeros(0, 784, 3, device=dev),
                   torch.eye(3, device=dev)[None], torch.zeros(1, 2, device=dev), torch.zeros(1, 2, device=dev), forward_exact_hessian=True)
    assert [tuple(t.shape) for t in e] == [(0,), (0, 1), (0, 3), (0, 4, 4), (0, 784)] and e[0].dtype == torch.bool
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(batch64, planar=False)
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    base = pnp_uncert(t(x2d), t(istd), t(x3d), t(K), t(ur), t(vr), z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=t(thr), inlier_opt_only=True)
    pose = torch.cat([base[1], base[2]], 1)
    v32, c32, h32 = exact_hessian_device(t(x2d), t(istd), t(x3d), t(K), t(ur), t(vr), pose, base[4].to(torch.uint8), base[0].to(torch.uint8), with_hessian=True)
    # fp16 storage: the kernel widens exactly, so it must equal the fp32 run on the fp16-rounded values
    h16 = lambda a: t(a, torch.float16)
    v16, c16, hh16 = exact_hessian_device(h16(x2d), h16(istd), h16(x3d), t(K), t(ur), t(vr), pose, base[4].to(torch.uint8), base[0].to(torch.uint8), with_hessian=True)
    v16b, c16b, hh16b = exact_hessian_device(h16(x2d).float(), h16(istd).float(), h16(x3d).float(), t(K), t(ur), t(vr), pose, base[4].to(torch.uint8),
                                            base[0].to(torch.uint8), with_hessian=True)
    assert torch.equal(hh16, hh16b) and torch.equal(c16, c16b) and torch.equal(v16, v16b)
    assert torch.equal(v32, base[0].to(torch.uint8))                 # no object of this batch has a singular exact Hessian
    lib = _lib.load()
    assert lib.mr_pnp_exact_hessian_batched(None, None, None, None, None, None, 0, None, 1, None, None, 1, None, None, 4, 16, 0.5, None, None, None, None) == -1
    assert lib.mr_pnp_exact_hessian_batched(None, None, None, None, None, None, 0, None, 1, None, None, 1, None, None, 0, 16, 0.5, None, None, None, None) == 0

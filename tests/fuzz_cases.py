"""Adversarial correspondence sets for the PnP boundary (numpy only: shared by the CPU tests, the `-m gpu` tests and tests/sweeps/gpu_epnp_fuzz.py).

make_case(mode, rng) -> [x2d (B,P,2), istd (B,P,2), x3d (B,P,3), K, u_range, v_range, ransac_thr] float32, a random half of the objects of a
synthetic batch damaged in the way `mode` names; the other half stays ordinary."""
import numpy as np

from monorun_amd import synthetic as syn

MODES = ('coincident', 'nan', 'zero_weights', 'overflow', 'planar', 'garbage', 'zero_threshold', 'inf', 'collinear', 'one_point')


def make_case(mode, rng, B=None, hw=None, planar_layout=None):
    """mode: one of MODES or its index.  B / hw / planar_layout (channel-planar strided views, as the pipeline hands them over) are drawn
    from rng when None."""
    if not isinstance(mode, str):
        mode = MODES[int(mode) % len(MODES)]
    B = int(rng.choice([1, 3, 64, 200])) if B is None else int(B)
    hw = int(rng.choice([3, 4, 8, 10, 28])) if hw is None else int(hw)
    b = syn.make_batch(B=B, hw=hw, seed=int(rng.integers(1 << 30)))
    planar_layout = bool(rng.integers(2)) if planar_layout is None else bool(planar_layout)
    x2d, istd, x3d, K, ur, vr, thr = [np.array(a, copy=True) for a in syn.pnp_boundary(b, planar=planar_layout)]
    P = x2d.shape[1]
    sel = rng.uniform(size=B) < 0.5
    if mode == 'coincident': x3d[sel] = 0.0                                   # all points coincide
    elif mode == 'nan': x2d[sel, rng.integers(P)] = np.nan                    # NaN correspondences
    elif mode == 'zero_weights': istd[sel] = 0.0
    elif mode == 'overflow': x3d[sel] *= 1e20
    elif mode == 'planar': x3d[sel, :, 1] = 0.0                               # planar object (rank-2 covariance: one control point collapses)
    elif mode == 'garbage': x3d[sel] = rng.normal(0, 1, x3d[sel].shape).astype(np.float32)
    elif mode == 'zero_threshold': thr[sel] = 0.0
    elif mode == 'inf': x2d[sel] = np.inf
    elif mode == 'collinear': x3d[sel, :, 0] = 0.0; x3d[sel, :, 1] = 0.0      # collinear object
    elif mode == 'one_point': x3d[sel] = x3d[sel][:, :1]                      # every point the same 3-D point, different pixels
    else: raise ValueError(mode)
    return [x2d, istd, x3d, K, ur, vr, thr]

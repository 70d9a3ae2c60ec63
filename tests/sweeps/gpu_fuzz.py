"""Adversarial inputs: the kernel must terminate, never report a NaN pose as valid, and agree with the oracle on validity
(development aid; run under `timeout`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from monorun_amd import synthetic as syn, _lib
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
from oracle import oracle as orc
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(os.environ.get('SEED', 0)))
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
bad = 0
for trial in range(int(os.environ.get('TRIALS', 40))):
    B = int(rng.choice([1, 3, 64, 200])); hw = int(rng.choice([2, 3, 8, 10, 28]))
    b = syn.make_batch(B=B, hw=hw, seed=int(rng.integers(1 << 30)))
    x2d, istd, x3d, K, ur, vr, thr = [np.array(a, copy=True) for a in syn.pnp_boundary(b, planar=bool(rng.integers(2)))]
    P = x2d.shape[1]
    mode = trial % 8
    sel = rng.uniform(size=B) < 0.5
    if mode == 0: x3d[sel] = 0.0                                   # all points coincide
    elif mode == 1: x2d[sel, rng.integers(P)] = np.nan             # NaN correspondences
    elif mode == 2: istd[sel] = 0.0                                # zero weights
    elif mode == 3: x3d[sel] *= 1e20                               # overflow
    elif mode == 4: istd[sel] = -istd[sel]                         # negative weights
    elif mode == 5: x3d[sel] = rng.normal(0, 1, x3d[sel].shape).astype(np.float32)   # garbage geometry
    elif mode == 6: thr[sel] = 0.0                                 # zero consensus threshold
    elif mode == 7: x2d[sel] = np.inf
    planar = x2d.strides[1] == 4
    for wpo in (0, 1, 2, 4):
        out = pnp_uncert_device(dv(x2d), dv(istd), dv(x3d), dv(K), dv(ur), dv(vr), 0.5, 0.6, dv(thr), True, flags=(wpo << _lib.MR_WAVES_SHIFT), with_diag=True)
        torch.cuda.synchronize()
        valid, pose = out[0].cpu().numpy().astype(bool), out[1].cpu().numpy()
        if not np.isfinite(pose[valid]).all():
            bad += 1; print('trial', trial, 'mode', mode, 'wpo', wpo, 'non-finite pose reported valid')
    with np.errstate(all='ignore'):
        o = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, num_threads=0)
    if not np.array_equal(valid, o[0]):
        d = np.flatnonzero(valid != o[0])
        print('trial', trial, 'mode', mode, 'B', B, 'P', P, 'validity differs from the oracle for', len(d), 'objects', d[:5])
        bad += 1
print('fuzz done, problems:', bad)

"""Kernel-time sweep over wavefronts-per-object and batch size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
for B in [int(x) for x in os.environ.get('QBS', '1024').split(',')]:
    b = syn.make_batch(B=B, seed=1234)
    x2d, istd, x3d, K, ur, vr, thr = [dv(a) for a in syn.pnp_boundary(b, planar=True)]
    for wpo in [int(x) for x in os.environ.get('QW', '0,1,2,3,4,8').split(',')]:
        L = PnPLaunch(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, flags=(wpo << 8))
        for _ in range(5): L.run()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for e0, e1 in ev:
            e0.record(); L.run(); e1.record()
        torch.cuda.synchronize()
        ms = np.array([e0.elapsed_time(e1) for e0, e1 in ev])
        print(f'B={B} wpo={wpo}: kernel {ms.mean()*1e3:.1f} us (min {ms.min()*1e3:.1f}), {B/ms.mean()*1e3:.3e} solves/s, valid {L.valid.float().mean().item():.3f}')

"""BASELINE config 5 shape on ONE GPU shard: 8192 objects x 56x56 correspondences, fp16 storage (development aid;
the 8-GPU run of the full 64k-object config belongs to the driver)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch
dev = torch.device('cuda:0')
B0 = 1024
b = syn.make_batch(B=B0, hw=56, seed=4321)
x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
rep = int(os.environ.get('REP', 8))
def planar_rep(a, dt):
    base = torch.from_numpy(np.ascontiguousarray(np.asarray(a).transpose(0, 2, 1))).to(dev).to(dt).repeat(rep, 1, 1)   # (B, C, P)
    return base.permute(0, 2, 1)
for dt, name, bps in ((torch.float16, 'fp16', 3136 * 7 * 2 + 56 + 85 + 3136), (torch.float32, 'fp32', 3136 * 7 * 4 + 56 + 85 + 3136)):
    X2, W, X3 = planar_rep(x2d, dt), planar_rep(istd, dt), planar_rep(x3d, dt)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    for wpo in (0, 2, 4, 8):
        L = PnPLaunch(X2, W, X3, t(K), t(ur), t(vr), 0.5, 0.6, t(thr).repeat(rep), True, flags=(wpo << 8))
        for _ in range(2): L.run()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for e0, e1 in ev:
            e0.record(); L.run(); e1.record()
        torch.cuda.synchronize()
        ms = np.mean([e0.elapsed_time(e1) for e0, e1 in ev])
        B = B0 * rep
        print(f'{name} B={B} P=3136 wpo={wpo}: {ms:.3f} ms/launch, {B/ms*1e3:.3e} solves/s, algorithmic {B*bps/ms/1e6:.1f} GB/s, valid {L.valid.float().mean().item():.3f}')

"""How long does the slowest object of the config-2 batch take when it is alone on the GPU? (development aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
dev = torch.device('cuda:0')
b = syn.make_batch(B=1024, seed=1234)
x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
out = pnp_uncert_device(dv(x2d), dv(istd), dv(x3d), dv(K), dv(ur), dv(vr), 0.5, 0.6, dv(thr), True, with_diag=True)
it = out[5][:, 0].cpu().numpy().astype(int)
order = np.argsort(-it)
print('LM iterations, top 8 objects:', [(int(i), int(it[i])) for i in order[:8]], ' median', np.median(it))
def time_subset(idx, wpo=4, reps=50):
    idx = np.asarray(idx)
    args = [dv(np.ascontiguousarray(a[idx].transpose(0, 2, 1)).transpose(0, 2, 1)) for a in (x2d, istd, x3d)]
    L = PnPLaunch(*args, dv(K), dv(ur), dv(vr), 0.5, 0.6, dv(thr[idx]), True, flags=(wpo << 8))
    for _ in range(5): L.run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); L.run(); e1.record()
    torch.cuda.synchronize()
    return np.mean([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3
for k in range(3):
    i = order[k]
    print(f'object {i} ({it[i]} iterations) alone: {time_subset([i]):.1f} us')
med = [i for i in range(1024) if it[i] == 3][:1]
print(f'a 3-iteration object alone: {time_subset(med):.1f} us')
keep = np.array([i for i in range(1024) if it[i] <= 9])
print(f'batch without the {1024 - len(keep)} objects above 9 iterations (B={len(keep)}): {time_subset(keep):.1f} us')
keep = np.array([i for i in range(1024) if it[i] <= 5])
print(f'batch without the {1024 - len(keep)} objects above 5 iterations (B={len(keep)}): {time_subset(keep):.1f} us')
print(f'full batch: {time_subset(np.arange(1024)):.1f} us')

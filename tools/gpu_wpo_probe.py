import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
def ev_time(L, reps=60, warm=5):
    for _ in range(warm): L.run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); L.run(); e1.record()
    torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3)
for seed in (1234, 1234 + 7919 * 11):
    b = syn.make_batch(B=1024, seed=seed)
    x = [dv(a) for a in syn.pnp_boundary(b, planar=True)]
    big = [torch.cat([a.permute(0, 2, 1).contiguous()] * 8, 0).permute(0, 2, 1) for a in x[:3]]
    for wpo in (1, 2, 3, 4, 8):
        L = PnPLaunch(x[0], x[1], x[2], x[3], x[4], x[5], 0.5, 0.6, x[6], True, flags=wpo << 8)
        L8 = PnPLaunch(big[0], big[1], big[2], x[3], x[4], x[5], 0.5, 0.6, x[6].repeat(8), True, flags=wpo << 8)
        print(f'seed {seed} wpo {wpo}: B=1024 {ev_time(L):.1f} us   B=8192 {ev_time(L8, 15):.1f} us')

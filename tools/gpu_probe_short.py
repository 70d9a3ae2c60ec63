"""Steady-state throughput of config-2 steps through PnPPipeline at depth 4, two waves per object (the bench's configuration), and of
isolated launches: the A/B figure for kernel variants that matter with launches in flight.  [MR_PNP_SO=variant.so] python tools/gpu_probe_short.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch, PnPPipeline
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
NB, S = 12, 24
batches = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234 + 7919 * i), planar=True)] for i in range(NB)]
tag = os.environ.get('MR_PNP_SO', 'default').split('/')[-1]
for waves, depth in [tuple(int(v) for v in w.split(":")) for w in os.environ.get("CASES", "2:4,0:1").split(",")]:
    ls = [[PnPLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, flags=(waves << 8)) for b in batches] for _ in range(S)]
    pipe = PnPPipeline(dev, depth=depth)
    for steps in (20, 480):
        res = []
        for rep in range(6):
            for i in range(12):
                pipe.submit(ls[i % S][i % NB], slot=i % S)
            pipe.drain(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                pipe.submit(ls[i % S][i % NB], slot=i % S)
            pipe.drain()
            res.append(1024 * steps / (time.perf_counter() - t0) / 1e6)
        print(f'{tag}: waves {waves} depth {depth} steps {steps}: ' + ' '.join(f'{r:6.2f}' for r in res) + f'  median {np.median(res):.2f} M solves/s', flush=True)
    chk = sum(float(l.pose.double().sum()) for l in ls[0])
    print(f'{tag}: pose checksum {chk:.9f}')
    del pipe, ls

"""How many launches in flight, and how many waves per object, once launches overlap?  (development aid)
Config-2 steps (1024 objects, 12 rotating batches) issued through monorun_amd.PnPPipeline at several depths / wave counts,
for a short window (20 steps: the driver's) and a long one.  GPU_MAX_HW_QUEUES (ROCclr: hardware queues the streams of a
process are mapped onto, default 4) is taken from the environment: run once with 4 and once with 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch, PnPPipeline
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
NB = 12
batches = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234 + 7919 * i), planar=True)] for i in range(NB)]
print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES', '(default)'))
print(f'{"waves":>5} {"depth":>5} {"steps":>5} ' + ' '.join(f'{"M/s #" + str(i):>8}' for i in range(4)))
S = 24
for waves in (0, 2):
    ls = [[PnPLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, flags=(waves << 8)) for b in batches] for _ in range(S)]
    for depth in (1, 2, 3, 4, 5, 6, 8):
        pipe = PnPPipeline(dev, depth=depth)
        for steps in (20, 240):
            res = []
            for rep in range(4):
                for i in range(5):
                    pipe.submit(ls[i % S][i % NB], slot=i % S)
                pipe.drain(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    pipe.submit(ls[i % S][i % NB], slot=i % S)
                pipe.drain()
                res.append(1024 * steps / (time.perf_counter() - t0) / 1e6)
            print(f'{waves:5d} {depth:5d} {steps:5d} ' + ' '.join(f'{r:8.2f}' for r in res), flush=True)
        del pipe
    del ls

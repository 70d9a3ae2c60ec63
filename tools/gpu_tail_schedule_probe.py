import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch, PnPPipeline
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
NB, S = 12, 24
batches = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234 + 7919 * i), planar=True)] for i in range(NB)]
mk = lambda w: [[PnPLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, flags=(w << 8)) for b in batches] for _ in range(S)]
l2, l4 = mk(2), mk(4)
pipe = PnPPipeline(dev, depth=4)
steps = 20
for rnd in range(3):
  head = 0
  for tail in (0, 2, 0, 2, 1, 3):
    res = []
    for rep in range(7):
        for i in range(5):
            pipe.submit(l2[i % S][i % NB], slot=i % S)
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            ls = l4 if (i >= steps - tail or i < head) else l2
            pipe.submit(ls[i % S][i % NB], slot=i % S)
        pipe.drain()
        res.append(1024 * steps / (time.perf_counter() - t0) / 1e6)
    print(f'first {head} / last {tail} launches with 4 waves: ' + ' '.join(f'{r:6.2f}' for r in res) + f'  median {sorted(res)[3]:6.2f}', flush=True)

"""The reference flow as the boundary runs it (the initialiser's launches + the LM launch that carries its re-fit), REPS times on config-2 batch 0 —
or, with STRESS=1, on the config-5 shard the bench's stress line runs (8192 x 56x56, fp16 storage) — for the profiler (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_epnp_device
dev = torch.device('cuda:0')
REPS = int(os.environ.get('REPS', 10))
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
if os.environ.get('STRESS') == '1':           # bench.py --workload stress: 1024 distinct objects tiled 8x, fp16 channel-planar
    npi = syn.pnp_boundary(syn.make_batch(B=1024, hw=56, seed=4321), planar=True)
    rep8 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).transpose(0, 2, 1))).to(dev).to(torch.float16).repeat(8, 1, 1).permute(0, 2, 1)
    x2d, istd, x3d = [rep8(a) for a in npi[:3]]
    K, ur, vr = [dv(a) for a in npi[3:6]]
    thr = dv(npi[6]).repeat(8)
else:
    x2d, istd, x3d, K, ur, vr, thr = [dv(a) for a in syn.pnp_boundary(syn.make_batch(B=int(os.environ.get('OBJECTS', 1024)), seed=1234), planar=True)]
for _ in range(REPS):
    pnp_uncert_epnp_device(x2d, istd, x3d, K, ur, vr, z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=thr, inlier_opt_only=True)
torch.cuda.synchronize()
print('ok', REPS)

"""One launch set of GROUP (default 5) calls (PnPEpnpGroupLaunch, the regime `value` is measured in: first round 3, two waves per object in the LM launch), REPS
times over config-2 batches 0..GROUP-1, for the profiler (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPEpnpLaunch, PnPEpnpGroupLaunch
dev = torch.device('cuda:0')
REPS = int(os.environ.get('REPS', 6)); G = int(os.environ.get('GROUP', 5))
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
bs = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234 + 7919 * i), planar=True)] for i in range(G)]
ls = [PnPEpnpLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, flags=2 << 8) for b in bs]
g = PnPEpnpGroupLaunch(ls)
for _ in range(REPS):
    g.run()
torch.cuda.synchronize()
print('ok', REPS, float(ls[0].pose.double().sum()))

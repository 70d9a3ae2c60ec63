"""Time of the 6-DoF refinement launch over 1024 config-2 objects (development aid).  [MR_PNP_SO=variant.so] python tools/gpu_pnp6_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device, pnp6_refine_device
dev = torch.device('cuda:0')
b = syn.make_batch(B=1024, seed=1234)
x = [torch.from_numpy(np.asarray(a)).to(dev) for a in syn.pnp_boundary(b, planar=False)]
x[0], x[1], x[2] = [t.permute(0, 2, 1).contiguous().permute(0, 2, 1) for t in x[:3]]
valid, pose, cov, tr, mask, _ = pnp_uncert_device(x[0], x[1], x[2], x[3], x[4], x[5], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=x[6], inlier_opt_only=True)
for _ in range(3): out = pnp6_refine_device(x[0], x[1], x[2], x[3], x[4], x[5], mask, pose, valid, z_min=0.5)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for e0, e1 in ev:
    e0.record(); out = pnp6_refine_device(x[0], x[1], x[2], x[3], x[4], x[5], mask, pose, valid, z_min=0.5); e1.record()
torch.cuda.synchronize()
t = np.array([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3
print(os.environ.get('MR_PNP_SO', 'default').split('/')[-1], 'pnp6 refine of 1024 objects: median %.1f us min %.1f' % (np.median(t), t.min()), 'valid', int(out[0].sum()), 'pose checksum %.9f' % float(out[1].double().sum()))

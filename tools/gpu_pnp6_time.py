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
print(os.environ.get('MR_PNP_SO', 'default').split('/')[-1], 'pnp6 refine of 1024 objects: median %.1f us min %.1f' % (np.median(t), t.min()), 'valid', int(out[0].sum()), 'pose checksum %.9f' % float(out[1].double().sum()), 'cov checksum %.9e' % float(out[2].double().abs().sum()))
# where the time goes: LM iterations of the refinement, and single objects alone on the GPU (the chain itself)
out = pnp6_refine_device(x[0], x[1], x[2], x[3], x[4], x[5], mask, pose, valid, z_min=0.5, with_diag=True)
it = out[3][:, 0].cpu().numpy().astype(int)
print('LM iterations of the 6-DoF refinement: mean %.2f median %d p99 %d max %d; histogram' % (it.mean(), np.median(it), np.percentile(it, 99), it.max()),
      {int(v): int(c) for v, c in zip(*np.unique(it, return_counts=True))})
def alone(idx, reps=30):
    sel = torch.as_tensor(np.asarray(idx), device=dev)
    a = [t[sel].permute(0, 2, 1).contiguous().permute(0, 2, 1) for t in x[:3]]
    m, p, v = mask[sel].contiguous(), pose[sel].contiguous(), valid[sel].contiguous()
    for _ in range(3): pnp6_refine_device(a[0], a[1], a[2], x[3], x[4], x[5], m, p, v, z_min=0.5)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); pnp6_refine_device(a[0], a[1], a[2], x[3], x[4], x[5], m, p, v, z_min=0.5); e1.record()
    torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])) * 1e3
order = np.argsort(-it)
for k in (0, 1):
    print('object %d (%d iterations) alone: %.1f us' % (order[k], it[order[k]], alone([order[k]])))
med = int(np.where(it == int(np.median(it)))[0][0])
print('object %d (%d iterations) alone: %.1f us' % (med, it[med], alone([med])))
for cap in (int(np.percentile(it, 99)), int(np.percentile(it, 90)), int(np.median(it))):
    keep = np.where(it <= cap)[0]
    print('batch without the %d objects above %d iterations: %.1f us' % (1024 - len(keep), cap, alone(keep)))
print('512 objects (one round of workgroups): %.1f us' % alone(np.arange(512)))

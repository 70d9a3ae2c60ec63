"""A/B of the initialiser's lane mappings (MR_EP_WIDE = 0 quad | 2 row | 4 wave per matrix — or MR_EP_WIDE_HYP / MR_EP_WIDE_BETAS for one launch —, read once per
process; unset = the library's rule): one call of the reference flow at a time, HIP-event time per call, and a digest of EVERY output so that the mappings can be compared bit for bit across processes.
    MR_EP_WIDE=4 OBJECTS=100,1024 python tools/gpu_wide_ab.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPEpnpLaunch
dev = torch.device('cuda:0')


def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d


NB = int(os.environ.get('NBATCH', 12))
for BO in [int(v) for v in os.environ.get('OBJECTS', '100,256,1024').split(',')]:
    batches = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=BO, seed=1234 + 7919 * i, outlier_frac=(0.15, 0.4)[i % 2]), planar=True)] for i in range(NB)]
    le = [PnPEpnpLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, with_diag=True) for b in batches]
    for l in le:
        l.run()
    torch.cuda.synchronize()
    per = []
    for rep in range(5):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in le]
        for l, (e0, e1) in zip(le, ev):
            e0.record(); l.run(); e1.record()
        torch.cuda.synchronize()
        per.append([e0.elapsed_time(e1) * 1e3 for e0, e1 in ev])
    per = np.median(np.array(per), axis=0)
    h = hashlib.sha256()
    for l in le:
        for t in (l.valid, l.pose, l.cov, l.tr, l.mask, l.init_pose, l.init_mask, l.init_valid, l.diag, l.init_diag):
            h.update(t.cpu().numpy().tobytes())
    print(f'hyp {os.environ.get("MR_EP_WIDE_HYP", os.environ.get("MR_EP_WIDE", "rule"))} betas {os.environ.get("MR_EP_WIDE_BETAS", os.environ.get("MR_EP_WIDE", "rule"))} objects {BO:5d}: us per call mean {per.mean():7.1f} min {per.min():7.1f} max {per.max():7.1f}; sha256 of all outputs {h.hexdigest()[:16]}', flush=True)

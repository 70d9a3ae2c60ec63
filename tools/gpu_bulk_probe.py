"""One large launch at a time (no launches in flight): candidate tile against full tile, two waves per object.  (development aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
print('library:', os.environ.get('MR_PNP_SO', '(default)'), ' MR_CT_PER_CU', os.environ.get('MR_CT_PER_CU'), ' MR_CT_NO_REDO', os.environ.get('MR_CT_NO_REDO'))
for B in (4096, 16384):
    b = [dv(a) for a in syn.pnp_boundary(syn.make_batch(B=B, seed=4321), planar=True)]
    for name, fl in (('candidate tile', 2 << 8), ('full tile', (2 << 8) | 0x80), ('candidate tile', 2 << 8), ('full tile', (2 << 8) | 0x80)):
        l = PnPLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, flags=fl)
        for _ in range(3): l.run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); l.run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f'B {B:6d} {name:15s}: median {ts[7]:8.1f} us  min {ts[0]:8.1f}  -> {B / ts[7]:6.2f} M solves/s; checksum {l.pose.double().sum().item():.9f}', flush=True)

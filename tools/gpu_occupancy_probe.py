"""What would more resident objects per CU buy the launches in flight?  (development aid)
The two-wave instantiation is LDS- and VGPR-bound at six objects per CU for 28x28 tiles.  Smaller tiles lift the LDS bound; a variant
build with the two-wave kernel squeezed to 128 VGPRs (tools/build_variant.sh sq128 -DMR_RELAXED_WPO=0) lifts the register bound.
Run once per library (MR_PNP_SO) and compare at equal tile size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch, PnPPipeline
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
NB, S = 12, 24
print('library:', os.environ.get('MR_PNP_SO', '(default)'))
for hw in [int(x) for x in os.environ.get("TILES", "28,24,22,20").split(",")]:
    batches = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, hw=hw, seed=1234 + 7919 * i), planar=True)] for i in range(NB)]
    ls = [[PnPLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, flags=(int(os.environ.get("WAVES", "2")) << 8) | int(os.environ.get("EXTRA_FLAGS", "0"), 0)) for b in batches] for _ in range(S)]
    pipe = PnPPipeline(dev, depth=4)
    for steps in (20, 240):
        res = []
        for rep in range(5):
            for i in range(5):
                pipe.submit(ls[i % S][i % NB], slot=i % S)
            pipe.drain(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                pipe.submit(ls[i % S][i % NB], slot=i % S)
            pipe.drain()
            res.append(1024 * steps / (time.perf_counter() - t0) / 1e6)
        chk = float(sum(l.pose.double().sum().item() for l in ls[0]))
        print(f'tile {hw}x{hw} steps {steps:3d}: ' + ' '.join(f'{r:6.2f}' for r in res) + f'  median {sorted(res)[2]:6.2f} M solves/s; pose checksum {chk:.9f}', flush=True)
    del pipe, ls, batches

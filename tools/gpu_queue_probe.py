"""Which HIP streams really run side by side?  (development aid)  (1) pairwise overlap of torch's pool streams, measured with
mr_spin; (2) config-2 launches with MR_ANY_ORDER on ONE stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch, _lib
dev = torch.device('cuda:0')
lib = _lib.load()
torch.zeros(1, device=dev); torch.cuda.synchronize()
cands = [('n%d' % i, torch.cuda.Stream(device=dev)) for i in range(10)] + [('h%d' % i, torch.cuda.Stream(device=dev, priority=-1)) for i in range(4)]
def pair_us(a, b, us=40):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.mr_spin(us, a.cuda_stream); lib.mr_spin(us, b.cuda_stream)
        a.synchronize(); b.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e6)
    return best
print('pairwise wall time of two 40 us spins (serialised ~ 90+, side by side ~ 50):')
print('      ' + ' '.join(f'{n:>5}' for n, _ in cands))
for i, (ni, si) in enumerate(cands):
    print(f'{ni:>5} ' + ' '.join(f'{pair_us(si, sj):5.0f}' if j > i else '    .' for j, (nj, sj) in enumerate(cands)), flush=True)

def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
NB = 12
batches = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234 + 7919 * i), planar=True)] for i in range(NB)]
for waves in (0, 2):
    for anyo in (0, 0x20):
        ls = [[PnPLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, flags=(waves << 8) | anyo) for b in batches] for _ in range(8)]
        for steps in (20, 240):
            res = []
            for rep in range(3):
                for i in range(5): ls[i % 8][i % NB].run()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(steps): ls[i % 8][i % NB].run()
                torch.cuda.synchronize()
                res.append(1024 * steps / (time.perf_counter() - t0) / 1e6)
            print(f'waves {waves} any_order {bool(anyo)} steps {steps}: ' + ' '.join(f'{r:6.2f}' for r in res) + ' M solves/s', flush=True)
        if anyo:
            ref = PnPLaunch(*batches[3][:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=batches[3][6], inlier_opt_only=True, flags=(waves << 8)); ref.run(); torch.cuda.synchronize()
            print('   results equal to an ordered launch:', bool(torch.equal(ref.pose, ls[3 % 8][3].pose) and torch.equal(ref.mask, ls[3 % 8][3].mask)))

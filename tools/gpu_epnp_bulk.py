"""The reference's flow (EPnP / RANSAC initialiser + LM) on ONE call over 8 x 1024 config-2 objects: results equal the 1024-object
calls', and the bulk rate (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPEpnpLaunch
dev = torch.device('cuda:0')
NB = int(os.environ.get('NB', 8))
bs = [[torch.from_numpy(np.asarray(a)).to(dev) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234 + 7919 * i), planar=False)] for i in range(NB)]
cat = lambda j: torch.cat([b[j] for b in bs], 0)
big = PnPEpnpLaunch(cat(0), cat(1), cat(2), bs[0][3], bs[0][4], bs[0][5], epnp_ransac_thres=cat(6), inlier_opt_only=True)
small = [PnPEpnpLaunch(*b[:6], epnp_ransac_thres=b[6], inlier_opt_only=True) for b in bs]
for l in small: l.run()
for _ in range(3): big.run()
torch.cuda.synchronize()
for i, l in enumerate(small):
    s = slice(1024 * i, 1024 * (i + 1))
    bad = [(n, int((getattr(big, n)[s] != getattr(l, n)).reshape(1024, -1).any(1).sum())) for n in ('init_pose', 'init_mask', 'init_valid', 'pose', 'mask', 'valid')]
    # (the covariance is compared to tolerance: the LM launch picks its waves-per-object by batch size, and the summation order of the
    #  covariance sums follows it — last-bit differences in float32)
    assert torch.allclose(big.cov[s], l.cov, rtol=1e-5, atol=0.0), i
    if any(c for _, c in bad):
        print('batch', i, 'objects that differ:', bad)
        if os.environ.get('STRICT', '1') == '1': raise SystemExit(1)
n = 10
t0 = time.perf_counter()
for _ in range(n): big.run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('one call over %d objects: %.3f ms = %.2f M solves/s; results equal the 1024-object calls; workspace %.0f MB' % (1024 * NB, dt * 1e3, 1024 * NB / dt / 1e6, big.work.numel() / 1e6))

"""Parity sweep over many seeds (development aid): counts objects that deviate from the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from monorun_amd import synthetic as syn
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
from oracle import oracle as orc
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
nseeds = int(os.environ.get('NSEEDS', 24)); B = 1024
tot = dict(objects=0, mask=0, iters=0, why=0, valid=0, pose=0, cov=0)
worst_pose = 0.0; worst_cov = 0.0
nthr = 16
for seed in range(100, 100 + nseeds):
    planar = seed % 2 == 0
    b = syn.make_batch(B=B, seed=seed, outlier_frac=[0.15, 0.3, 0.05][seed % 3], noise_3d=[0.03, 0.08][seed % 2])
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=planar)
    ref = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, num_threads=nthr)
    out = pnp_uncert_device(dv(x2d), dv(istd), dv(x3d), dv(K), dv(ur), dv(vr), 0.5, 0.6, dv(thr), True, with_diag=True)
    torch.cuda.synchronize()
    valid, pose, cov, tr, mask, diag = [t.cpu().numpy() for t in out]
    mm = (mask.astype(bool) != ref[5]).any(1)
    it = diag[:, 0] != ref[6][:, 0]; wy = diag[:, 2] != ref[6][:, 2]; vv = valid.astype(bool) != ref[0]
    dp = np.maximum(np.abs(np.angle(np.exp(1j * (pose[:, 0] - ref[1][:, 0])))), np.abs(pose[:, 1:] - ref[2]).max(1))
    ok = ref[0] & valid.astype(bool)
    sc = np.abs(ref[3]).reshape(B, -1).max(1)
    dc = np.abs(cov - ref[3]).reshape(B, -1).max(1) / sc
    tot['objects'] += B; tot['mask'] += mm.sum(); tot['iters'] += it.sum(); tot['why'] += wy.sum(); tot['valid'] += vv.sum()
    tot['pose'] += (dp[ok] > 1e-4).sum(); tot['cov'] += (dc[ok] > 1e-5).sum()
    worst_pose = max(worst_pose, dp[ok].max()); worst_cov = max(worst_cov, dc[ok].max())
    print(f'seed {seed} planar={planar}: valid {ref[0].mean():.3f} mask-mismatch objs {mm.sum()} iter-mismatch {it.sum()} max|dpose| {dp[ok].max():.2e} max rel dcov {dc[ok].max():.2e} iters max {int(ref[6][:,0].max())}')
print('TOTAL', tot, 'worst pose', worst_pose, 'worst cov', worst_cov)

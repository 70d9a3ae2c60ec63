"""Why does the LM launch behind the EPnP initialiser take longer than the whole fused launch?  LM iteration
histograms and isolated launch times (HIP events) of: the fused launch (mask + K0 + LM + covariance), the from-init launch fed by the
EPnP initialiser, and the from-init launch fed by K0's own result (same kernel instantiation, K0's starts).  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn
from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device, pnp_uncert_from_init_device, pnp_uncert_device
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return np.median(ts), min(ts)
for seed in (1234, 1234 + 7919):
    x2d, istd, x3d, K, ur, vr, thr = [dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=seed), planar=True)]
    ini, im, iv, _, _ = epnp_ransac_device(x2d, istd, x3d, K, epnp_istd_thres=0.6, epnp_ransac_thres=thr)
    fused = pnp_uncert_device(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, with_diag=True)
    fi = pnp_uncert_from_init_device(x2d, istd, x3d, K, ur, vr, ini, im, iv, z_min=0.5, inlier_opt_only=True, with_diag=True)
    torch.cuda.synchronize()
    it_f, it_e = fused[5][:, 0].cpu().numpy(), fi[5][:, 0].cpu().numpy()
    print(f'seed {seed}: LM iterations fused (K0 starts) mean {it_f.mean():.2f} max {it_f.max():.0f}; from EPnP starts mean {it_e.mean():.2f} max {it_e.max():.0f}; '
          f'inliers fused {fused[4].sum().item() / 1024:.1f} epnp {fi[4].sum().item() / 1024:.1f}')
    print('   histogram K0  :', np.bincount(it_f.astype(int), minlength=12)[:40].tolist())
    print('   histogram EPnP:', np.bincount(it_e.astype(int), minlength=12)[:40].tolist())
    # K0's result as an external initialiser: pose f64 + mask from the fused launch
    k0_pose = fused[1].double().contiguous(); k0_mask = fused[4].contiguous(); k0_valid = fused[0].contiguous()
    t_f = timed(lambda: pnp_uncert_device(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True))
    t_e = timed(lambda: pnp_uncert_from_init_device(x2d, istd, x3d, K, ur, vr, ini, im, iv, z_min=0.5, inlier_opt_only=True))
    t_k = timed(lambda: pnp_uncert_from_init_device(x2d, istd, x3d, K, ur, vr, k0_pose, k0_mask, k0_valid, z_min=0.5, inlier_opt_only=True))
    t_i = timed(lambda: epnp_ransac_device(x2d, istd, x3d, K, epnp_istd_thres=0.6, epnp_ransac_thres=thr))
    print(f'   us (median, min): fused {t_f[0]:.1f} {t_f[1]:.1f} | from-init(EPnP) {t_e[0]:.1f} {t_e[1]:.1f} | from-init(K0 final pose+mask) {t_k[0]:.1f} {t_k[1]:.1f} | initialiser {t_i[0]:.1f} {t_i[1]:.1f}')

"""Parity sweep of the EPnP / RANSAC initialiser path (pnp_uncert(..., initialiser='epnp') = mr_epnp_ransac_batched + mr_pnp_uncert_from_init_batched)
over many seeds against the reference's flow restated (oracle u2d_pnp_epnp): counts objects that deviate.
    NSEEDS=200 python tests/sweeps/gpu_epnp_parity_sweep.py                  # config-2 shape: 1024 objects x 28x28 per seed
    HW=56 B=256 NSEEDS=8 python tests/sweeps/gpu_epnp_parity_sweep.py       # config-5 shape; odd pairs of seeds store fp16
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from monorun_amd import synthetic as syn
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_epnp_device
from oracle import oracle as orc
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
nseeds = int(os.environ.get('NSEEDS', 24)); B = int(os.environ.get('B', 1024)); HW = int(os.environ.get('HW', 28))
tot = dict(objects=0, init_valid=0, ransac_mask=0, init_pose=0, lm_iters=0, lm_why=0, valid=0, pose=0, cov=0)
worst_init = worst_pose = worst_cov = 0.0
nthr = 16
t_gpu = t_cpu = 0.0
for seed in range(100, 100 + nseeds):
    planar = seed % 2 == 0
    b = syn.make_batch(B=B, hw=HW, seed=seed, outlier_frac=[0.15, 0.3, 0.05, 0.45][seed % 4], noise_3d=[0.03, 0.08][seed % 2])
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=planar)
    half = HW != 28 and (seed // 2) % 2 == 1
    if half:
        keep = lambda a: np.ascontiguousarray(a.astype(np.float16).astype(np.float32).transpose(0, 2, 1)).transpose(0, 2, 1) if planar else \
            np.ascontiguousarray(a.astype(np.float16).astype(np.float32))
        x2d, istd, x3d = keep(x2d), keep(istd), keep(x3d)
    t0 = time.perf_counter()
    ref = orc.u2d_pnp_epnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, return_init=True, num_threads=nthr)
    t_cpu += time.perf_counter() - t0
    h = (lambda t: t.to(torch.float16)) if half else (lambda t: t)
    d = [h(dv(x2d)), h(dv(istd)), h(dv(x3d)), dv(K), dv(ur), dv(vr), dv(thr)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    # the boundary's default path: the initialiser's six launches + the LM launch that carries its re-fit (mr_pnp_uncert_from_epnp_grouped)
    out = pnp_uncert_epnp_device(d[0], d[1], d[2], d[3], d[4], d[5], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=d[6], inlier_opt_only=True, with_diag=True,
                                 flags=int(os.environ.get('WPO', 0)) << 8)           # WPO: waves per object of the LM launch (0 = the library's choice)
    torch.cuda.synchronize(); t_gpu += time.perf_counter() - t0
    ini, iv = out[6], out[7]
    valid, pose, cov, tr, mask, diag = [t.cpu().numpy() for t in out[:6]]
    r_ret, r_yaw, r_t, r_cov, r_tr, r_mask, r_diag, r_init = ref
    init_ok = r_diag[:, 2] != 8                       # 8 = the initialiser failed
    mm = (mask.astype(bool) != r_mask).any(1)
    di = np.abs(ini.cpu().numpy() - r_init).max(1)
    it = diag[:, 0] != r_diag[:, 0]; wy = diag[:, 2] % 16 != r_diag[:, 2]; vv = valid.astype(bool) != r_ret
    dp = np.maximum(np.abs(np.angle(np.exp(1j * (pose[:, 0] - r_yaw[:, 0])))), np.abs(pose[:, 1:] - r_t).max(1))
    ok = r_ret & valid.astype(bool)
    sc = np.abs(r_cov).reshape(B, -1).max(1)
    dc = np.abs(cov - r_cov).reshape(B, -1).max(1) / sc
    tot['objects'] += B; tot['init_valid'] += int((iv.cpu().numpy().astype(bool) != init_ok).sum()); tot['ransac_mask'] += int(mm.sum())
    tot['init_pose'] += int((di > 1e-9).sum()); tot['lm_iters'] += int(it.sum()); tot['lm_why'] += int(wy.sum()); tot['valid'] += int(vv.sum())
    tot['pose'] += int((dp[ok] > 1e-4).sum()); tot['cov'] += int((dc[ok] > 1e-5).sum())
    worst_init = max(worst_init, float(di.max())); worst_pose = max(worst_pose, float(dp[ok].max())); worst_cov = max(worst_cov, float(dc[ok].max()))
    print(f'seed {seed} planar={planar} fp16={half} outliers {[0.15, 0.3, 0.05, 0.45][seed % 4]}: valid {r_ret.mean():.3f} init-failed {int((~init_ok).sum())} mask-mismatch objs {int(mm.sum())} '
          f'max|dinit| {di.max():.1e} lm-iter-mismatch {int(it.sum())} max|dpose| {dp[ok].max():.2e} max rel dcov {dc[ok].max():.2e}', flush=True)
print('TOTAL deviating objects', tot, 'worst init pose', worst_init, 'worst pose', worst_pose, 'worst cov', worst_cov)
print(f'time: GPU path {t_gpu:.2f} s, CPU restatement ({nthr} threads) {t_cpu:.2f} s for {tot["objects"]} objects')

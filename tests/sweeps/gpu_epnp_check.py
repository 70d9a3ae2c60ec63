"""GPU EPnP/RANSAC initialiser against the CPU restatement, object by object (development sweep; B, SEED, HW from the environment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from monorun_amd import synthetic as syn
from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device, pnp_uncert_from_init_device
from oracle import oracle as orc
B, SEED, HW = int(os.environ.get('B', 64)), int(os.environ.get('SEED', 7)), int(os.environ.get('HW', 28))
dev = torch.device('cuda:0')
b = syn.make_batch(B=B, hw=HW, seed=SEED)
x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=False)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t0 = time.perf_counter()
ini, imask, ivalid, diag, hyp = epnp_ransac_device(t(x2d), t(istd), t(x3d), t(K), epnp_istd_thres=0.6, epnp_ransac_thres=t(thr), with_diag=True, debug_hypotheses=True)
torch.cuda.synchronize()
print('gpu call', time.perf_counter() - t0)
ini, imask, ivalid, diag, hyp = [a.cpu().numpy() for a in (ini, imask, ivalid, diag, hyp)]
cand = orc.istd_inlier_mask(istd, np.float32(0.6))
bad = dict(hyp=0, cnt=0, mask=0, ok=0, pose=0, iters=0)
worst_h, worst_p = 0.0, 0.0
for i in range(B):
    m = cand[i] if cand[i].sum() > 4 else np.ones_like(cand[i])
    idx = np.nonzero(m)[0]
    r = orc.epnp_ransac_trace(x3d[i][idx], x2d[i][idx], K.reshape(-1, 9)[0], float(thr[i]))
    ev = r['cnt'] >= 0
    dh = np.abs(hyp[i][ev] - r['hyp'][ev])
    dh = np.where(np.isnan(hyp[i][ev]) & np.isnan(r['hyp'][ev]), 0.0, dh)
    worst_h = max(worst_h, np.nanmax(dh) if dh.size else 0.0)
    bad['hyp'] += int(not np.all(dh <= 1e-9))
    full = np.zeros_like(m); 
    if r['ok']:
        full[idx] = r['mask']
    else:
        full = m
    bad['mask'] += int(not np.array_equal(full.astype(np.uint8), imask[i]))
    bad['ok'] += int(bool(r['ok']) != bool(ivalid[i]))
    bad['iters'] += int(r['iters'] != int(diag[i, 0]))
    if r['ok']:
        ref = np.array([r['rvec'][1], *r['tvec']])
        dp = np.abs(ref - ini[i]).max()
        worst_p = max(worst_p, dp)
        bad['pose'] += int(not dp <= 1e-9)
    if i < 3:
        print(i, 'n', len(idx), 'iters', r['iters'], int(diag[i, 0]), 'best', int(r['mask'].sum()), int(diag[i, 1]), 'hyp diff', np.nanmax(dh) if dh.size else None, 'pose', ini[i], ref if r['ok'] else None)
print('objects', B, 'mismatches', bad, 'worst hypothesis diff', worst_h, 'worst init pose diff', worst_p)
# end to end
ref = orc.u2d_pnp_epnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True)
ini_t, im_t, iv_t, _, _ = epnp_ransac_device(t(x2d), t(istd), t(x3d), t(K), epnp_istd_thres=0.6, epnp_ransac_thres=t(thr))
valid, pose, cov, tr, mask, dg = pnp_uncert_from_init_device(t(x2d), t(istd), t(x3d), t(K), t(ur), t(vr), ini_t, im_t, iv_t, z_min=0.5, inlier_opt_only=True, with_diag=True)
torch.cuda.synchronize()
valid, pose, mask = valid.cpu().numpy().astype(bool), pose.cpu().numpy(), mask.cpu().numpy().astype(bool)
print('end to end: valid equal', np.array_equal(valid, ref[0]), 'mask equal', np.array_equal(mask, ref[5]),
      'yaw', np.abs(np.angle(np.exp(1j * (pose[:, :1] - ref[1]))))[ref[0]].max(), 't', np.abs(pose[:, 1:] - ref[2])[ref[0]].max())
# timing: the initialiser launch and the LM launch behind it
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
dx = [t(a) for a in (x2d, istd, x3d, K, ur, vr, thr)]
for rep in range(3):
    e[0].record()
    ini_t, im_t, iv_t, _, _ = epnp_ransac_device(dx[0], dx[1], dx[2], dx[3], epnp_istd_thres=0.6, epnp_ransac_thres=dx[6])
    e[1].record()
    pnp_uncert_from_init_device(dx[0], dx[1], dx[2], dx[3], dx[4], dx[5], ini_t, im_t, iv_t, z_min=0.5, inlier_opt_only=True)
    e[2].record()
    torch.cuda.synchronize()
    print(f'B = {B}: EPnP/RANSAC launch {e[0].elapsed_time(e[1]) * 1e3:.0f} us, LM launch {e[1].elapsed_time(e[2]) * 1e3:.0f} us')
# per-stage times: tools/profile_epnp_quick.sh (rocprofv3 kernel trace of the initialiser's launches)

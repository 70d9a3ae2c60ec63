"""Quick GPU sanity + timing sweep (development aid; the judged numbers come from bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from monorun_amd import synthetic as syn, _lib
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
from oracle import oracle as orc

dev = torch.device('cuda:0')
print(torch.cuda.get_device_name(0))
B = int(os.environ.get('QB', 256))
b = syn.make_batch(B=B, seed=1234)
for planar in (True, False):
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=planar)
    o = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, num_threads=0)
    def dv(a):
        t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
    args = (dv(x2d), dv(istd), dv(x3d), dv(K), dv(ur), dv(vr))
    for wpo in (1, 2, 4, 8):
        valid, pose, cov, tr, mask, diag = pnp_uncert_device(*args, z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=dv(thr),
                                                             inlier_opt_only=True, flags=(wpo << _lib.MR_WAVES_SHIFT), with_diag=True)
        torch.cuda.synchronize()
        valid, pose, cov, tr, mask, diag = [t.cpu().numpy() for t in (valid, pose, cov, tr, mask, diag)]
        dpose = np.abs(pose - np.concatenate([o[1], o[2]], 1))
        relcov = np.abs(cov - o[3]).reshape(B, -1).max(1) / np.abs(o[3]).reshape(B, -1).max(1)
        print(f'planar={planar} wpo={wpo}: valid eq {np.array_equal(valid.astype(bool), o[0])}, mask mismatches {(mask.astype(bool) != o[5]).sum()} '
              f'(objects {((mask.astype(bool) != o[5]).any(1)).sum()}), max|dpose| {dpose.max(0)}, cov rel max {relcov.max():.2e}, '
              f'iters eq {np.array_equal(diag[:, 0], o[6][:, 0])}, why eq {np.array_equal(diag[:, 2] % 16, o[6][:, 2])}, tr eq {np.allclose(tr, o[4][:, 0], rtol=1e-6)}')
# timing
b = syn.make_batch(B=1024, seed=1234)
x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
args = (dv(x2d), dv(istd), dv(x3d), dv(K), dv(ur), dv(vr)); thr_d = dv(thr)
for wpo in (0, 1, 2, 4, 8):
    fl = (wpo << _lib.MR_WAVES_SHIFT)
    for _ in range(5):
        pnp_uncert_device(*args, 0.5, 0.6, thr_d, True, flags=fl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        pnp_uncert_device(*args, 0.5, 0.6, thr_d, True, flags=fl)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f'B=1024 wpo={wpo}: {ms*1e3:.1f} us/step, {1024/ms*1e3:.3e} solves/s, algorithmic {1024*22877/ms/1e6:.1f} GB/s')

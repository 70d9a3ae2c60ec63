"""Cycle stamps of the hypothesis kernel's phases (lane 0 of the first 64 workgroups, round 1), a -DMR_DEBUG_STAMPS build:
   tools/build_variant.sh stamps -DMR_DEBUG_STAMPS;  MR_PNP_SO=monorun_amd/variants/libmr_stamps.so python tools/gpu_hyp_timeline.py
Development aid (profiles/r05_hyp_timeline.txt)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, _lib
from monorun_amd.ops.least_squares.pnp_uncert import epnp_ransac_device
dev = torch.device('cuda:0'); lib = _lib.load(); lib.mr_pnp_debug_set_stamps.argtypes = [ctypes.c_void_p]
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
x2d, istd, x3d, K, ur, vr, thr = [dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234 + 7919), planar=True)]   # a batch without a second round
st = torch.zeros(64, 24, dtype=torch.int64, device=dev)
for it in range(3):
    st.zero_()
    lib.mr_pnp_debug_set_stamps(st.data_ptr())
    epnp_ransac_device(x2d, istd, x3d, K, epnp_istd_thres=0.6, epnp_ransac_thres=thr)
    torch.cuda.synchronize()
lib.mr_pnp_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64)
names = [(0, 1, 'entry -> samples loaded, points'), (1, 2, 'control points (3x3 Jacobi eig) + barycentric'), (2, 3, 'park + M^T M rows'),
         (3, 10, 'eig12: tridiagonalisation'), (10, 11, 'eig12: scaling + 56 bisection steps'), (11, 12, 'eig12: LU + 3 inverse iterations'),
         (12, 13, 'eig12: unit vectors + Gram-Schmidt'), (13, 4, 'eig12: back-transformation'), (4, 5, 'broadcast + L, rho'),
         (5, 6, 'beta approximation (6x5 Jacobi SVD)'), (6, 7, 'Gauss-Newton (5 x QR)'), (7, 8, 'pose (3x3 SVD) + reprojection error')]
tot = np.median(s[:, 8] - s[:, 0])
print(f'hypothesis kernel, lane 0 of 64 workgroups, median cycles (total {tot:.0f}):')
for a, b, n in names:
    d = s[:, b] - s[:, a]
    print(f'  {n:50s} {np.median(d):9.0f}  ({100 * np.median(d) / tot:4.1f} %)   min {d.min():.0f} max {d.max():.0f}')

"""Per-stage cycle breakdown of the fused kernel (development aid)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, _lib
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
dev = torch.device('cuda:0')
lib = _lib.load()
lib.mr_pnp_debug_set_stamps.argtypes = [ctypes.c_void_p]
B = int(os.environ.get('QB', 1024))
b = syn.make_batch(B=B, seed=1234)
x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
args = (dv(x2d), dv(istd), dv(x3d), dv(K), dv(ur), dv(vr)); thr_d = dv(thr)
names = ['load', 'mask', 'k0:list+hyp', 'k0:consensus', 'k0:refit', 'lm', 'cov']
for wpo in (1, 2, 4):
    st = torch.zeros(B, 24, dtype=torch.int64, device=dev)
    for it in range(3):
        lib.mr_pnp_debug_set_stamps(st.data_ptr())
        out = pnp_uncert_device(*args, 0.5, 0.6, thr_d, True, flags=(wpo << _lib.MR_WAVES_SHIFT), with_diag=True)
        torch.cuda.synchronize()
    lib.mr_pnp_debug_set_stamps(None)
    s = st.cpu().numpy().astype(np.float64)
    wall = (s[:, 9] - s[:, 8]) * 10.0   # ns
    dbg = s[:, 12:24]
    print('   LM detail (median cycles): eval0 sincos %d points %d reduce %d | post-eval0 -> iter1 start %d | solve+logic before eval1 %d | eval1 sincos %d points %d reduce %d | after-eval logic %d' % tuple(np.median(x) for x in (dbg[:,1]-dbg[:,0], dbg[:,2]-dbg[:,1], dbg[:,3]-dbg[:,2], dbg[:,4]-dbg[:,3], dbg[:,5]-dbg[:,4], dbg[:,7]-dbg[:,6], dbg[:,8]-dbg[:,7], dbg[:,9]-dbg[:,8], dbg[:,10]-dbg[:,9])))
    s = s[:, :8]
    d = np.diff(s, axis=1)
    print('   block wall time median %.1f us, max %.1f us; s_memtime ticks per ns: %.3f' % (np.median(wall)/1e3, wall.max()/1e3, np.median((s[:,7]-s[:,0]) / wall)))
    iters = out[5][:, 0].cpu().numpy()
    print(f'wpo={wpo}  median cycles per stage (100 MHz constant clock ticks -> x24 for 2.4GHz shader cycles?)')
    for n, col in zip(names, d.T):
        print(f'   {n:14s} median {np.median(col):10.0f}  mean {col.mean():10.0f}  max {col.max():10.0f}')
    print('   total', np.median(s[:, 7] - s[:, 0]), ' span of whole grid', s[:, 7].max() - s[:, 0].min(), 'LM iters mean', iters.mean(),
          ' lm ticks per (iter+1):', np.median(d[:, 5] / (iters + 1)))

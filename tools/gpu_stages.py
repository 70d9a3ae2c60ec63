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
    st = torch.zeros(B, 10, dtype=torch.int64, device=dev)
    for it in range(3):
        lib.mr_pnp_debug_set_stamps(st.data_ptr())
        out = pnp_uncert_device(*args, 0.5, 0.6, thr_d, True, flags=(wpo << _lib.MR_WAVES_SHIFT), with_diag=True)
        torch.cuda.synchronize()
    lib.mr_pnp_debug_set_stamps(None)
    s = st.cpu().numpy().astype(np.float64)
    hw = s[:, 8].astype(np.int64); xcc = s[:, 9].astype(np.int64) & 0xf; s = s[:, :8]
    d = np.diff(s, axis=1)
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
    key = xcc * 10000 + se * 1000 + sh * 100 + cu
    import collections
    cnt = collections.Counter(key.tolist())
    print('   distinct CUs used', len(cnt), 'blocks per CU histogram', collections.Counter(cnt.values()), 'simd hist', collections.Counter(simd.tolist()))
    for x in range(8):
        m = xcc == x
        if m.any(): print('   xcc', x, 'blocks', m.sum(), 'span cycles', s[m, 7].max() - s[m, 0].min(), 'start spread', s[m, 0].max() - s[m, 0].min())
    iters = out[5][:, 0].cpu().numpy()
    print(f'wpo={wpo}  median cycles per stage (100 MHz constant clock ticks -> x24 for 2.4GHz shader cycles?)')
    for n, col in zip(names, d.T):
        print(f'   {n:14s} median {np.median(col):10.0f}  mean {col.mean():10.0f}  max {col.max():10.0f}')
    print('   total', np.median(s[:, 7] - s[:, 0]), ' span of whole grid', s[:, 7].max() - s[:, 0].min(), 'LM iters mean', iters.mean(),
          ' lm ticks per (iter+1):', np.median(d[:, 5] / (iters + 1)))

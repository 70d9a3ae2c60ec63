"""Where does the leader wave of an object spend one LM pass?  (stands in for a thread trace: the rocprofv3 ATT decoder is not in the image)

Stamps build (tools/build_variant.sh stamps -DMR_DEBUG_STAMPS; run with MR_PNP_SO=monorun_amd/variants/libmr_stamps.so): thread 0 of every
object reads the shader cycle counter at the phase boundaries of the first evaluation and of LM iteration 1, and at the stage boundaries
of the kernel.  Objects are measured ALONE on the GPU (nothing else competes for the SIMD: the chain itself) and inside the full batch.
Every stamp costs the leader an s_memtime + a store (~40 - 60 cycles); the release build carries none of this.
"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, _lib
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
dev = torch.device('cuda:0')
lib = _lib.load()
lib.mr_pnp_debug_set_stamps.argtypes = [ctypes.c_void_p]
b = syn.make_batch(B=1024, seed=1234)
x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
def sub(idx):
    idx = np.asarray(idx)
    pts = [dv(np.ascontiguousarray(a[idx].transpose(0, 2, 1)).transpose(0, 2, 1)) for a in (x2d, istd, x3d)]
    return pts + [dv(K), dv(ur), dv(vr)], dv(thr[idx])
full, thr_full = sub(np.arange(1024))
out = pnp_uncert_device(*full, 0.5, 0.6, thr_full, True, with_diag=True)
iters = out[5][:, 0].cpu().numpy().astype(int)
n_inl = out[4].sum(1).cpu().numpy().astype(int)
order = np.argsort(-iters)
tail, typical = int(order[0]), int(np.where(iters == 3)[0][0])
names = ['load', 'istd mask', 'K0 list + hypotheses', 'K0 consensus', 'K0 refit', 'LM', 'covariance + outputs']
def run(idx, wpo, label):
    args, t = sub(idx)
    B = len(idx)
    st = torch.zeros(B, 24, dtype=torch.int64, device=dev)
    for _ in range(4):
        lib.mr_pnp_debug_set_stamps(st.data_ptr())
        o = pnp_uncert_device(*args, 0.5, 0.6, t, True, flags=(wpo << _lib.MR_WAVES_SHIFT), with_diag=True)
        torch.cuda.synchronize()
    lib.mr_pnp_debug_set_stamps(None)
    s = st.cpu().numpy().astype(np.float64)
    it = o[5][:, 0].cpu().numpy().astype(int)
    ni = o[4].sum(1).cpu().numpy().astype(int)
    wall_ns = (s[:, 9] - s[:, 8]) * 10.0
    ghz = np.median((s[:, 7] - s[:, 0]) / np.maximum(wall_ns, 1.0))
    d = np.diff(s[:, :8], axis=1)
    g = s[:, 12:24]
    med = lambda x: float(np.median(x))
    print(f'--- {label}: {B} object(s), {wpo} waves per object; LM iterations {int(np.median(it))} (median) / {it.max()} (max), inliers {int(np.median(ni))}; '
          f'shader clock {ghz:.2f} GHz; workgroup lifetime median {med(wall_ns) / 1e3:.1f} us, max {wall_ns.max() / 1e3:.1f} us')
    print('    kernel stages, cycles (us) of thread 0:')
    for n, col in zip(names, d.T):
        print(f'      {n:22s} {med(col):9.0f}  ({med(col) / ghz / 1e3:6.2f} us)')
    lm_c = d[:, 5]
    per_pass = med(lm_c / (it + 1))
    print(f'      LM / (iterations + 1)  {per_pass:9.0f}  ({per_pass / ghz / 1e3:6.2f} us per pass)')
    ph = [('first evaluation: sincos + message + barrier', g[:, 1] - g[:, 0]), ('first evaluation: point loop', g[:, 2] - g[:, 1]),
          ('first evaluation: block reduction', g[:, 3] - g[:, 2]), ('set-up between evaluation 0 and iteration 1 (Jacobi scale, gradient test)', g[:, 4] - g[:, 3]),
          ('iteration 1: 4x4 damped solve + step logic', g[:, 5] - g[:, 4]), ('iteration 1: sincos + message + barrier', g[:, 7] - g[:, 6]),
          ('iteration 1: point loop', g[:, 8] - g[:, 7]), ('iteration 1: block reduction', g[:, 9] - g[:, 8]),
          ('iteration 1: tolerance tests (parameter, function), relative decrease', g[:, 10] - g[:, 9]),
          ('iteration 1: step acceptance, radius update, gradient test of the next pass', g[:, 11] - g[:, 10])]
    print('    leader wave, phases of one LM pass, cycles (us):')
    tot = 0.0
    for n, col in ph:
        ok = it >= (2 if 'acceptance' in n else 1)
        v = med(col[ok]) if ok.any() else float('nan')
        print(f'      {n:78s} {v:7.0f}  ({v / ghz / 1e3:5.2f} us)')
        if n.startswith('iteration 1'):
            tot += v
    print(f'      {"iteration 1, sum of its phases":78s} {tot:7.0f}  ({tot / ghz / 1e3:5.2f} us)')
print(f'batch 0 of config 2: tail object {tail} ({iters[tail]} LM iterations, {n_inl[tail]} inliers), typical object {typical} ({iters[typical]} iterations, {n_inl[typical]} inliers)')
for wpo in (4, 2):
    run([tail], wpo, 'the tail object alone on the GPU')
    run([typical], wpo, 'a 3-iteration object alone on the GPU')
    run(np.arange(1024), wpo, 'the full 1024-object batch (medians over objects)')

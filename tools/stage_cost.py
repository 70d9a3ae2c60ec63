"""Per-stage cost of the fused kernel from early-exit builds (development aid).
  for k in 1..7: tools/build_variant.sh exit$k -DMR_EXIT_AFTER=$k   (stops after stage k: 1 load, 2 mask+list, 3 hypotheses, 4 consensus, 5 refit, 6 LM, 7 cov)
  then on the GPU box:  python tools/stage_cost.py   -> kernel time of every variant at B = 1 (object 0 alone), 100 (one image), 1024 and 8192 (differences = stage costs)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
from monorun_amd import synthetic as syn, PnPLaunch
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
b = syn.make_batch(B=1024, seed=1234)
x = [dv(a) for a in syn.pnp_boundary(b, planar=True)]
def t(L, reps=40):
    for _ in range(5): L.run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); L.run(); e1.record()
    torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3)
L = PnPLaunch(x[0], x[1], x[2], x[3], x[4], x[5], 0.5, 0.6, x[6], True)
big = [torch.cat([a.permute(0, 2, 1).contiguous()] * 8, 0).permute(0, 2, 1) for a in x[:3]]
L8 = PnPLaunch(big[0], big[1], big[2], x[3], x[4], x[5], 0.5, 0.6, x[6].repeat(8), True)
def sub(n, o=0):
    xs = [a[o:o + n].permute(0, 2, 1).contiguous().permute(0, 2, 1) for a in x[:3]]
    return PnPLaunch(xs[0], xs[1], xs[2], x[3], x[4], x[5], 0.5, 0.6, x[6][o:o + n].contiguous(), True)
print('%%.1f %%.1f %%.1f %%.1f' %% (t(sub(1, 1)), t(sub(100)), t(L), t(L8, 15)))
''' % ROOT
prev = (0.0, 0.0, 0.0, 0.0)
names = ['', 'load', 'mask+list', 'hypotheses', 'consensus', 'refit', 'LM', 'cov+out']
for k in list(range(1, 8)) + ['full']:
    so = os.path.join(ROOT, 'monorun_amd', 'variants', f'libmr_exit{k}.so' if k != 'full' else 'libmr_full.so')
    if not os.path.exists(so):
        continue
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MR_PNP_SO=so), capture_output=True, text=True)
    try:
        a1, a100, a, b8 = [float(v) for v in r.stdout.split()[-4:]]
    except Exception:
        print(k, 'failed', r.stderr[-500:]); continue
    print(f'exit after {k} ({names[k] if k != "full" else "full kernel"}):  B=1 {a1:6.1f} us (+{a1 - prev[2]:5.1f})  B=100 {a100:6.1f} us (+{a100 - prev[3]:5.1f})  B=1024 {a:7.1f} us (+{a - prev[0]:5.1f})   B=8192 {b8:7.1f} us (+{b8 - prev[1]:6.1f}, {(b8 - prev[1]) / 8:5.1f} per 1024)')
    prev = (a, b8, a1, a100)

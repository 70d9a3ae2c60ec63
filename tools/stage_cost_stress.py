"""Per-stage cost of the fused kernel on the config-5 shard shape (8192 objects x 56x56, fp16 storage) from early-exit builds
(development aid; build the variants as for tools/stage_cost.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
from monorun_amd import synthetic as syn, PnPLaunch
dev = torch.device('cuda:0')
b = syn.make_batch(B=1024, hw=56, seed=4321)
x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=True)
rep = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).transpose(0, 2, 1))).to(dev).to(torch.float16).repeat(8, 1, 1).permute(0, 2, 1)
t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
L = PnPLaunch(rep(x2d), rep(istd), rep(x3d), t(K), t(ur), t(vr), 0.5, 0.6, t(thr).repeat(8), True)
for _ in range(3): L.run()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for e0, e1 in ev:
    e0.record(); L.run(); e1.record()
torch.cuda.synchronize()
print('%%.1f' %% float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3))
''' % ROOT
prev = 0.0
names = ['', 'load', 'mask+list', 'hypotheses', 'consensus', 'refit', 'LM', 'cov+out']
for k in list(range(1, 8)) + ['full']:
    so = os.path.join(ROOT, 'monorun_amd', 'variants', f'libmr_exit{k}.so' if k != 'full' else 'libmr_full.so')
    if not os.path.exists(so):
        continue
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MR_PNP_SO=so), capture_output=True, text=True)
    try:
        a = float(r.stdout.split()[-1])
    except Exception:
        print(k, 'failed', r.stderr[-300:]); continue
    print(f'exit after {k} ({names[k] if k != "full" else "full kernel"}): {a:8.1f} us (+{a - prev:7.1f})')
    prev = a

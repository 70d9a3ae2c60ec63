"""Kernel performance probe used while optimising (development aid): one line per regime.
    [MR_PNP_SO=path/to/variant.so] python tools/gpu_kperf.py [tag]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch, _lib
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get('MR_PNP_SO', 'default')
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
def ev_time(L, reps=60, warm=5):
    for _ in range(warm): L.run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); L.run(); e1.record()
    torch.cuda.synchronize()
    t = np.array([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3
    return float(np.median(t)), float(t.min())
batches = []
for s in (1234, 1234 + 7919 * 11):          # batch 0 and the batch with the 39-iteration object
    b = syn.make_batch(B=1024, seed=s)
    batches.append([dv(a) for a in syn.pnp_boundary(b, planar=True)])
def mk(x, idx=None, wpo=0, **kw):
    x2d, istd, x3d, K, ur, vr, thr = x
    if idx is not None:
        sel = lambda a: a.permute(0, 2, 1)[idx].contiguous().permute(0, 2, 1)
        x2d, istd, x3d, thr = sel(x2d), sel(istd), sel(x3d), thr[idx]
    return PnPLaunch(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, flags=(wpo << 8), **kw)
out = {}
L0 = mk(batches[0], with_diag=True); L0.run(); torch.cuda.synchronize()
it = L0.diag[:, 0].cpu().numpy().astype(int)
order = np.argsort(-it)
out['b0_1024'] = ev_time(mk(batches[0]))
out['b11_1024'] = ev_time(mk(batches[1]))
big = [torch.cat([a.permute(0, 2, 1).contiguous()] * 8, 0).permute(0, 2, 1) for a in batches[0][:3]] + batches[0][3:6] + [batches[0][6].repeat(8)]
out['b0x8_8192'] = ev_time(mk(big), reps=20)
out['slow_obj_alone(%d it)' % it[order[0]]] = ev_time(mk(batches[0], torch.tensor([int(order[0])], device=dev)))
i3 = int(np.where(it == 3)[0][0])
out['3it_obj_alone'] = ev_time(mk(batches[0], torch.tensor([i3], device=dev)))
out['first100'] = ev_time(mk(batches[0], torch.arange(100, device=dev)))
keep = torch.tensor(np.where(it <= 5)[0], device=dev)
out['b0_le5it(%d)' % len(keep)] = ev_time(mk(batches[0], keep))
# the headline regime: 12 rotating batches, 4 launches in flight, the pipeline's wave count
if os.environ.get('KPERF_PIPE', '1') == '1':
    from monorun_amd import PnPPipeline
    nb = [batches[0]] + [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234 + 7919 * i), planar=True)] for i in range(1, 11)] + [batches[1]]
    pipe = PnPPipeline(dev, depth=4)
    fl = pipe.flags_for(1024, 784)
    ls = [[PnPLaunch(*b[:6], 0.5, 0.6, b[6], True, flags=fl) for b in nb] for _ in range(8)]
    res = []
    for rep in range(5):
        for i in range(12): pipe.submit(ls[i % 8][i % 12], slot=i % 8)
        pipe.drain(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(240): pipe.submit(ls[i % 8][i % 12], slot=i % 8)
        pipe.drain()
        res.append((time.perf_counter() - t0) / 240 * 1e6)
    out['pipe4_us_per_step'] = (float(np.median(res)), float(min(res)))
print(f'[{tag}] ' + '  '.join(f'{k}={v[0]:.1f}/{v[1]:.1f}' for k, v in out.items()) + '  (median/min us)')

import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch
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
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3)
b = syn.make_batch(B=1024, seed=1234)
x = [dv(a) for a in syn.pnp_boundary(b, planar=True)]
L0 = PnPLaunch(x[0], x[1], x[2], x[3], x[4], x[5], 0.5, 0.6, x[6], True, with_diag=True); L0.run(); torch.cuda.synchronize()
it = L0.diag[:, 0].cpu().numpy().astype(int); order = np.argsort(-it)
for idx, name in ((order[0], '17-it'), (int(np.where(it == 3)[0][0]), '3-it')):
    sel = lambda a: a.permute(0, 2, 1)[idx:idx+1].contiguous().permute(0, 2, 1)
    for wpo in (2, 3, 4, 8):
        L = PnPLaunch(sel(x[0]), sel(x[1]), sel(x[2]), x[3], x[4], x[5], 0.5, 0.6, x[6][idx:idx+1], True, flags=wpo << 8)
        print(name, 'wpo', wpo, '%.1f us' % ev_time(L), 'n_inl', int(L0.mask[idx].sum()))

"""Per-wave timeline of one K2 launch (development aid; needs a -DMR_K2_EXPERIMENT build selected with MR_PNP_SO): 100 MHz wall-clock
stamps of every wave at 0 start, 1 object parameters ready, 2 pixel loads issued, 3 pixel data arrived, 4 arithmetic done (last quad),
5 stores issued, 6 stores complete."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, _lib
from monorun_amd.pose_head import NocDecodeLaunch
dev = torch.device('cuda:0')
lib = _lib.load()
lib.mr_pnp_debug_set_stamps.argtypes = [ctypes.c_void_p]
k2s = []
for i in range(3):
    b = syn.make_batch(B=1024, hw=28, seed=1234 + 7919 * i)
    all_pred, dim = syn.encode_head_outputs(b, seed=1234 + i)
    rng = np.random.default_rng(1234 + i)
    dim_var = torch.from_numpy((0.01 * rng.random((1024, 3)) + 1e-4).astype(np.float32)).to(dev)
    ap, lab, dm, rois = [torch.from_numpy(x).to(dev) for x in (all_pred, b['labels'], dim, b['rois'])]
    k2s.append(NocDecodeLaunch(ap, lab, False, dm, dim_var, rois))
thr = int(os.environ.get('MR_K2_THREADS', '256'))
nw = 1024 * (thr // 64)
st = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
for r in range(12): k2s[r % 3].run()
torch.cuda.synchronize()
lib.mr_pnp_debug_set_stamps(st.data_ptr())
k2s[0].run()
torch.cuda.synchronize()
lib.mr_pnp_debug_set_stamps(None)
s = st.cpu().numpy().reshape(nw, 8).astype(np.float64)
s = s[s[:, 6] > 0]          # waves that ran the quad path (an object's last wave, in pair mode, leaves before the later stamps)
nw = len(s)
t0 = s[:, 0].min()
s = (s - t0) / 100.0          # us
names = ['start', 'params ready', 'loads issued', 'data arrived', 'arithmetic done', 'stores issued', 'stores complete']
print(f'threads {thr}, LDS cap {os.environ.get("MR_K2_LDS", "0")}: {nw} waves; us since the first wave started')
for i, n in enumerate(names):
    v = s[:, i]
    print(f'  {n:<16} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}')
d = np.diff(s[:, :7], axis=1)
for i in range(6):
    print(f'  {names[i]} -> {names[i + 1]}: median {np.median(d[:, i]):6.2f}  p10 {np.percentile(d[:, i], 10):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f}')
full = s[:, 6].max()
busy = lambda a, b: sum(((s[:, a] <= t) & (s[:, b] > t)).sum() for t in np.arange(0, full, 0.25)) / max(1, len(np.arange(0, full, 0.25)))
print(f'  waves waiting for pixel data (avg over the launch) {busy(2, 3):7.0f}, in arithmetic {busy(3, 4):7.0f}, waiting for stores {busy(5, 6):7.0f}')

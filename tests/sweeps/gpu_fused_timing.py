"""Head output -> pose: one fused launch vs K2 + PnP (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import numpy as np, torch
from monorun_amd import synthetic as syn
from monorun_amd.pose_head import UncertPropPnPOptimizer, pose_from_head
from oracle import oracle as orc
dev = torch.device('cuda:0')
for B in (100, 1024):
    b = syn.make_batch(B=B, seed=99)
    rng = np.random.default_rng(5)
    labels, flip = b['labels'], rng.uniform(size=B) < 0.5
    mu, sd = orc.DIM_MEANS[labels], orc.DIM_STDS[labels]
    dim = ((b['dims'] - mu) / sd).astype(np.float32); dim_var = (rng.uniform(0.02, 0.1, (B, 3)) ** 2).astype(np.float32)
    dims, _ = orc.dim_decode(dim, dim_var, labels)
    noc = ((b['coords_3d'] / dims[:, :, None, None] - orc.NOC_MEANS[:, None, None]) / orc.NOC_STDS[:, None, None]).astype(np.float32)
    all_pred = rng.normal(0, 1, (B, 30, 28, 28)).astype(np.float32)
    _, _, chan = orc.slice_pred(all_pred, labels, flip)
    ar = np.arange(B)
    for k in range(3): all_pred[ar, chan[:, k]] = noc[:, k]
    for k in range(2): all_pred[ar, chan[:, 3 + k]] = (b['logstd'][:, k] - np.log(2.0)).astype(np.float32) * 0.5
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    args = (t(all_pred), t(labels), t(flip), t(dim), t(dim_var), t(b['rois']), t(b['K']), b['img_shape'])
    head = UncertPropPnPOptimizer().to(dev)
    for fused in (True, False):
        with torch.no_grad():
            for _ in range(5): pose_from_head(head, *args, fused=fused)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); c0 = time.perf_counter()
            for _ in range(30): r = pose_from_head(head, *args, fused=fused)
            c1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
        print(f'B={B} fused={fused}: {e0.elapsed_time(e1)/30*1e3:.1f} us per head->pose call (includes the torch glue ops; host-side issue time {(c1-c0)/30*1e6:.1f} us), valid {r["ret_val"].float().mean().item():.3f}')

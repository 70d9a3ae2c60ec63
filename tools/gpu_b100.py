"""B = 100 per-image regime: kernel time by waves-per-object, and the host-side cost of the three ways to issue it (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn
from monorun_amd.pose_head import UncertPropPnPOptimizer, PoseFromHeadLaunch
dev = torch.device('cuda:0')
b = syn.make_batch(B=100, seed=1234)
all_pred, dim = syn.encode_head_outputs(b, seed=1)
t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
head = UncertPropPnPOptimizer().to(dev)
for wpo in (0, 2, 4, 8):
    L = PoseFromHeadLaunch(head, t(all_pred), t(b['labels']), False, t(dim), None, t(b['rois']), t(b['K']), (375, 1242), flags=wpo << 8)
    for _ in range(5): L.run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for e0, e1 in ev:
        e0.record(); L.run(); e1.record()
    torch.cuda.synchronize()
    k = np.median([e0.elapsed_time(e1) for e0, e1 in ev]) * 1e3
    L.capture()
    res = {}
    for name, fn in (('run', L.run), ('replay', L.replay)):
        t1 = time.perf_counter()
        for _ in range(300):
            fn(); torch.cuda.current_stream().synchronize()
        res[name] = (time.perf_counter() - t1) / 300 * 1e6
    print(f'wpo {wpo}: kernel (events) {k:.1f} us   wall per synced call: prepared {res["run"]:.1f} us, graph replay {res["replay"]:.1f} us   valid {int(L.out["ret_val"].sum())}')

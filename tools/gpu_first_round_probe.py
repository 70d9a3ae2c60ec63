"""One call of the reference flow at a time, by batch size and by `first_round` (hypotheses solved for every object before the replayed RANSAC
loop is consulted; 30 = no second round at all): HIP-event time of a prepared PnPEpnpLaunch, median over the batches.  Result-neutral: the pose
checksum is printed.  [OBJECTS=100,256,1024] [FIRST=3,8,10,12,16,30] [MR_PNP_SO=variant.so] python tools/gpu_first_round_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPEpnpLaunch
dev = torch.device('cuda:0')


def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d


NB = int(os.environ.get('NBATCH', 12))
for BO in [int(v) for v in os.environ.get('OBJECTS', '100,256,1024').split(',')]:
    batches = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=BO, seed=1234 + 7919 * i), planar=True)] for i in range(NB)]
    for fr in [v for v in os.environ.get('FIRST', '0,3,8,10,12,16,30').split(',')]:
        fr = int(fr)
        le = [PnPEpnpLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, first_round=fr or None) for b in batches]
        for l in le:
            l.run()
        torch.cuda.synchronize()
        per = []
        for rep in range(5):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in le]
            for l, (e0, e1) in zip(le, ev):
                e0.record(); l.run(); e1.record()
            torch.cuda.synchronize()
            per.append([e0.elapsed_time(e1) * 1e3 for e0, e1 in ev])
        per = np.median(np.array(per), axis=0)
        print(f'objects {BO:5d} first_round {fr if fr else "rule":>4}: us per call over {NB} batches: mean {per.mean():7.1f} min {per.min():7.1f} max {per.max():7.1f}  '
              f'-> {BO / per.mean():.3f} M solves/s; pose checksum {sum(float(l.pose.double().sum()) for l in le):.9f}', flush=True)
        del le

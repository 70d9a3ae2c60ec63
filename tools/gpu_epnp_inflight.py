"""The reference-flow step (EPnP / RANSAC initialiser + LM) one call at a time and with four PnPEpnpLaunch objects in flight: the A/B
figure for changes to the initialiser's launches.  [MR_PNP_SO=variant.so] python tools/gpu_epnp_inflight.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPEpnpLaunch, PnPEpnpGroupLaunch, PnPPipeline
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
NB = int(os.environ.get('NBATCH', 12)); BO = int(os.environ.get('OBJECTS', 1024))      # OBJECTS=2048: what two calls grouped into one launch set would cost
batches = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=BO, seed=1234 + 7919 * i), planar=True)] for i in range(NB)]
tag = os.environ.get('MR_PNP_SO', 'default').split('/')[-1] + f' objects={BO} group={os.environ.get("GROUP", 1)} first_round=' + os.environ.get('MR_EPNP_FIRST_ROUND', '8')
for depth in [int(v) for v in os.environ.get('DEPTHS', '1,4').split(',')]:
    pipe = PnPPipeline(dev, depth=depth, record_events=False)
    le = [PnPEpnpLaunch(*batches[i % NB][:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=batches[i % NB][6], inlier_opt_only=True,
                        flags=(int(os.environ['LMWPO']) << 8) if 'LMWPO' in os.environ else (pipe.flags_for(BO, 784) if depth > 1 else 0)) for i in range(max(depth, NB))]
    G = int(os.environ.get('GROUP', 1))                # calls per launch set (PnPEpnpGroupLaunch)
    calls = le
    if G > 1:
        le = [PnPEpnpGroupLaunch(calls[k:k + G]) for k in range(0, len(calls) - G + 1, G)]
    res = []; host = []
    for rep in range(5):
        for i in range(8):
            pipe.submit(le[i % len(le)], slot=i % len(le))
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(96):
            pipe.submit(le[i % len(le)], slot=i % len(le))
        host.append((time.perf_counter() - t0) / 96 * 1e6)
        pipe.drain()
        res.append(BO * G * 96 / (time.perf_counter() - t0) / 1e6)
    print(f'{tag}: depth {depth} (streams found {pipe.depth}, {pipe.overlap_test}): ' + ' '.join(f'{r:5.2f}' for r in res) + f'  median {np.median(res):.2f} M solves/s; host {np.median(host):.0f} us per submit of {np.median([BO * G / r for r in res]):.0f}; pose checksum {sum(float(l.pose.double().sum()) for l in calls):.9f}', flush=True)
    del pipe, le

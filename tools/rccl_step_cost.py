"""What does the per-step exchange cost on the compute stream? (development aid; world size 1)
Back-to-back 1024-object launches with, per step: nothing | an event record | record + side-stream wait + a 90 KB device copy there |
record + side-stream wait + ncclAllGather there (private communicator) | the same on the compute stream itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from monorun_amd import synthetic as syn, PnPLaunch
from monorun_amd.parallel import PackedResults, RcclAllGather
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
NB, R = 6, 8
batches = [[dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234 + 7919 * i), planar=True)] for i in range(NB)]
packs = [PackedResults(1024, dev) for _ in range(R)]
masks = [torch.empty(1024, 784, device=dev, dtype=torch.uint8) for _ in range(R)]
WPO = int(os.environ.get('WPO', 0))
L = [[PnPLaunch(*b[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=b[6], inlier_opt_only=True, out=packs[k], mask=masks[k], flags=(WPO << 8)) for k in range(R)] for b in batches]
recv = [torch.empty_like(p.buf) for p in packs]
ag = RcclAllGather(dev)
side = torch.cuda.Stream(device=dev)
hi = torch.cuda.Stream(device=dev, priority=-1)
hi2 = torch.cuda.Stream(device=dev, priority=-1)
dones = [torch.cuda.Event() for _ in range(R)]
def run(mode, steps=200):
    dn = [None] * R
    ev = torch.cuda.Event()
    for w in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % R
            if mode == 'wait_batched' and i % (R // 2) == 0:
                for j in range(R // 2):
                    kk = (i + j) % R
                    if dn[kk] is not None:
                        torch.cuda.current_stream().wait_event(dn[kk])
            L[i % NB][k].run()
            if mode == 'wait_batched':
                dn[k] = ag.gather(packs[k].buf, recv[k])
            if mode == 'record':
                ev.record()
            elif mode == 'side_copy':
                ev.record(); side.wait_event(ev)
                with torch.cuda.stream(side):
                    recv[k].copy_(packs[k].buf, non_blocking=True)
            elif mode == 'side_rccl':
                ag.gather(packs[k].buf, recv[k])
            elif mode == 'side_rccl_noevent':
                ev.record(); side.wait_event(ev)
                ag.lib.ncclAllGather(packs[k].buf.data_ptr(), recv[k].data_ptr(), packs[k].buf.numel(), 1, ag.comm, side.cuda_stream)
            elif mode == 'side_rccl_prealloc':
                ev.record(); side.wait_event(ev)
                ag.lib.ncclAllGather(packs[k].buf.data_ptr(), recv[k].data_ptr(), packs[k].buf.numel(), 1, ag.comm, side.cuda_stream)
                dones[k].record(side)
            elif mode == 'side_rccl_query':
                if i >= R and not dn[k].query():
                    torch.cuda.current_stream().wait_event(dn[k])
                dn[k] = ag.gather(packs[k].buf, recv[k])
            elif mode == 'inline_agstream':
                ev.record(); ag.stream.wait_event(ev)
                ag.lib.ncclAllGather(packs[k].buf.data_ptr(), recv[k].data_ptr(), packs[k].buf.numel(), 1, ag.comm, ag.stream.cuda_stream)
                dones[k].record(ag.stream)
            elif mode == 'inline_devctx':
                ev.record(); side.wait_event(ev)
                with torch.cuda.device(dev):
                    ag.lib.ncclAllGather(packs[k].buf.data_ptr(), recv[k].data_ptr(), packs[k].buf.numel(), 1, ag.comm, side.cuda_stream)
                dones[k].record(side)
            elif mode == 'inline_agready':
                ag._ready.record(torch.cuda.current_stream(dev)); side.wait_event(ag._ready)
                ag.lib.ncclAllGather(packs[k].buf.data_ptr(), recv[k].data_ptr(), packs[k].buf.numel(), 1, ag.comm, side.cuda_stream)
                dones[k].record(side)
            elif mode in ('inline_hiprio', 'inline_hiprio2'):
                hs = hi if mode == 'inline_hiprio' else hi2
                ev.record(); hs.wait_event(ev)
                ag.lib.ncclAllGather(packs[k].buf.data_ptr(), recv[k].data_ptr(), packs[k].buf.numel(), 1, ag.comm, hs.cuda_stream)
                dones[k].record(hs)
            elif mode == 'side_rccl_waitalways':
                if dn[k] is not None:
                    torch.cuda.current_stream().wait_event(dn[k])
                dn[k] = ag.gather(packs[k].buf, recv[k])
            elif mode == 'wait_batched':
                pass
            elif mode == 'same_stream_rccl':
                ag.lib.ncclAllGather(packs[k].buf.data_ptr(), recv[k].data_ptr(), packs[k].buf.numel(), 1, ag.comm, torch.cuda.current_stream().cuda_stream)
        issue = (time.perf_counter() - t0) / steps * 1e6
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps * 1e6
    return el, issue
for m in ('none', 'record', 'side_copy', 'side_rccl', 'side_rccl_noevent', 'side_rccl_prealloc', 'side_rccl_query', 'side_rccl_waitalways', 'wait_batched', 'same_stream_rccl', 'none'):
    el, issue = run(m)
    print(f'wpo={WPO} {m:18s} {el:7.1f} us/step  (host issue {issue:5.1f} us/step)')
ag.close(); dist.destroy_process_group()

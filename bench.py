#!/usr/bin/env python3
"""bench.py — PnP solves/second on BASELINE.json's config 2 (1024 proposals x 28x28 correspondences).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic batch that is already resident in HBM:
the fused HIP kernel (istd mask -> K0 initialiser -> LM -> covariance, through the C ABI) over
1024 objects per GPU, plus — when N > 1 — the single RCCL all-gather of the packed per-object results
(north_star: "objects shard across the GPUs with an RCCL all-gather of poses").  Weak scaling: every
rank owns 1024 objects.  W untimed warm-up steps, then EXACTLY K steps between barrier +
torch.cuda.synchronize() pairs; the reported time is the MAX over ranks; rank 0 prints ONE JSON line.

Extra objects in the line (prompt ④):
  roofline      dominant kernel's algorithmic bytes / its average launch duration (HIP events around
                each launch, on the stream it is launched on), against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (C restatement of the reference's path, kind "port": the reference's own
                C++ needs Ceres and cannot be built here) timed on this box's host cores on a bounded
                sample of the same workload; rank 0, N = 1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from monorun_amd import synthetic as syn  # noqa: E402
from monorun_amd import PnPLaunch  # noqa: E402
from monorun_amd.parallel import PackedResults, ROW_BYTES  # noqa: E402

B_PER_GPU = 1024
HW = 28
P = HW * HW
SEED = 1234
# in = P*(2+2+3)*4 + 36 + 16 + 4 ; out = 16 + 64 + 4 + 1 + P  ->  22 877 B / solve (fp32, P = 784)
BYTES_PER_SOLVE = P * 7 * 4 + 36 + 16 + 4 + 16 + 64 + 4 + 1 + P
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary (multi-stream / 8192-object) throughput figures')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target CPU-baseline sample time')
    ap.add_argument('--waves', type=int, default=0, help='wavefronts per object (0 = library heuristic)')
    ap.add_argument('--workload', choices=['config2', 'stress'], default='config2',
                    help="config2 (default, the metric's configuration) or stress = BASELINE config 5's per-GPU shard: 8192 objects x "
                         '56x56 correspondences, fp16 storage (a parity-test shape; an extra line, never the judged one)')
    return ap.parse_args()


def to_dev(a, dev):
    t = torch.from_numpy(np.asarray(a))
    d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev)
    d.copy_(t)
    return d


def cpu_baseline(np_inputs, seconds):
    """Oracle (C restatement, fp64, -O2 like the reference) on the same workload: 1 thread = the
    reference's execution model (serial multi_apply, Ceres num_threads=1); all cores = fair ceiling."""
    from oracle import oracle as orc
    x2d, istd, x3d, K, ur, vr, thr = np_inputs
    n1 = 256                                           # bounded sample: first 256 objects of the batch, repeated
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.u2d_pnp(x2d[:n1], istd[:n1], x3d[:n1], K, ur, vr, 0.5, 0.6, thr[:n1], True, num_threads=1)
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds * 0.6 or reps >= 200:
            break
    one = dict(value=n1 * reps / el, unit='solves/s', cores=1, kind='port',
               sample=f'first {n1} objects of the config-2 batch x {reps} repeats, {el:.1f} s, single thread '
                      '(the reference runs objects serially with Ceres num_threads=1)')
    nthr = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                               # containers: honour the cgroup CPU quota (e.g. "1600000 100000" = 16 cores)
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            nthr = max(1, min(nthr, int(int(q) / int(per))))
    except Exception:  # noqa: BLE001
        pass
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, num_threads=nthr)
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds * 0.4 or reps >= 200:
            break
    allc = dict(value=x2d.shape[0] * reps / el, unit='solves/s', cores=nthr, kind='port',
                sample=f'full {x2d.shape[0]}-object batch x {reps} repeats, {el:.1f} s, OpenMP over objects')
    return one, allc


def main():
    global B_PER_GPU, HW, P, SEED, BYTES_PER_SOLVE
    args = parse()
    stress = args.workload == 'stress'
    if stress:                                   # 47 181 B / solve (fp16, P = 3136)
        B_PER_GPU, HW, SEED = 8192, 56, 4321
        P = HW * HW
        BYTES_PER_SOLVE = P * 7 * 2 + 36 + 16 + 4 + 16 + 64 + 4 + 1 + P
        args.no_cpu_baseline = args.no_secondary = True
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N > 1')
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # MR_BENCH_FORCE_DIST=1 runs the RCCL path (init, all-gather, barrier, max-reduce) even at world size 1 — the only way
    # to exercise it on a 1-GPU box
    use_dist = world > 1 or os.environ.get('MR_BENCH_FORCE_DIST') == '1'
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    # synthetic config-2 batch for this rank (different objects per rank: seed + rank), resident in HBM
    if stress:                                   # 1024 distinct objects, tiled 8x (generation time), stored as fp16 channel-planar
        batch = syn.make_batch(B=1024, hw=HW, seed=SEED + rank)
        np_inputs = syn.pnp_boundary(batch, planar=True)
        rep8 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).transpose(0, 2, 1))).to(dev).to(torch.float16).repeat(8, 1, 1).permute(0, 2, 1)
        x2d, istd, x3d = [rep8(a) for a in np_inputs[:3]]
        K, ur, vr = [to_dev(a, dev) for a in np_inputs[3:6]]
        thr = to_dev(np_inputs[6], dev).repeat(8)
    else:
        batch = syn.make_batch(B=B_PER_GPU, hw=HW, seed=SEED + rank)
        np_inputs = syn.pnp_boundary(batch, planar=True)     # the strided views the reference's head hands to the PnP
        x2d, istd, x3d, K, ur, vr, thr = [to_dev(a, dev) for a in np_inputs]
    # two result buffers: with N > 1 the all-gather of step i overlaps the kernel of step i+1
    packs = [PackedResults(B_PER_GPU, dev) for _ in range(2)]
    launches = [PnPLaunch(x2d, istd, x3d, K, ur, vr, z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=thr,
                          inlier_opt_only=True, flags=(args.waves << 8), out=pk) for pk in packs]
    packed, launch = packs[0], launches[0]
    gathered = [torch.empty(world * pk.buf.numel(), dtype=torch.uint8, device=dev) for pk in packs] if use_dist else None
    # exchange: a private RCCL communicator driven directly (ncclAllGather on a side stream, ~5 us of host time per call);
    # MR_BENCH_COMM=torch, or any failure to set it up, falls back to torch.distributed's (synchronous) all-gather
    rccl = None
    if use_dist and os.environ.get('MR_BENCH_COMM', 'rccl') == 'rccl':
        try:
            from monorun_amd.parallel import RcclAllGather
            rccl = RcclAllGather(dev)
        except Exception as e:                                   # noqa: BLE001 — any setup problem: use the c10d path
            print(f'[bench] direct RCCL path unavailable ({e}); using torch.distributed', file=sys.stderr)
            rccl = None
    done = [None, None]
    counter = [0]

    def step():
        if not use_dist:
            launch.run()
            return
        k = counter[0] & 1
        counter[0] += 1
        if rccl is None:
            launches[k].run()
            dist.all_gather_into_tensor(gathered[k], packs[k].buf)
            return
        if done[k] is not None:
            torch.cuda.current_stream().wait_event(done[k])     # the gather that read buffer k finished before it is rewritten
        launches[k].run()
        done[k] = rccl.gather(packs[k].buf, gathered[k])

    def fence():
        if use_dist:
            for k in range(2):
                if done[k] is not None:
                    torch.cuda.current_stream().wait_event(done[k])
                    done[k] = None
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant-kernel duration: HIP events around each launch on the launch stream (torch's current stream)
    n_ev = min(args.steps, 200)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    torch.cuda.synchronize()
    for e0, e1 in evs:
        e0.record()
        launch.run()
        e1.record()
    torch.cuda.synchronize()
    k_ms = np.array([e0.elapsed_time(e1) for e0, e1 in evs])
    kernel_ms = float(k_ms.mean())
    valid_frac = float(packed.valid.float().mean().item())
    assert valid_frac > 0.95, f'only {valid_frac:.3f} of the solves are valid — refusing to report a number'

    # Secondary, clearly labelled throughput figures (never `value`):
    #  (a) the same K steps issued round-robin on 4 HIP streams (independent batches in flight, as a serving loop
    #      would): at B = 1024 a launch lasts as long as its slowest object, streams fill the idle SIMDs of the tail;
    #  (b) one launch over 8 such batches (8192 objects): the kernel's throughput regime.
    extra = {}
    if world == 1 and not args.no_secondary:
        n_str = 4
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_str)]
        launches = [PnPLaunch(x2d, istd, x3d, K, ur, vr, z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=thr,
                              inlier_opt_only=True, flags=(args.waves << 8)) for _ in range(n_str)]
        for i in range(2 * n_str):
            launches[i % n_str].run(streams[i % n_str].cuda_stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            launches[i % n_str].run(streams[i % n_str].cuda_stream)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        extra['pipelined_4_streams'] = {'value': B_PER_GPU * args.steps / el, 'unit': 'solves/s', 'ms_per_step': el / args.steps * 1e3}
        # keep the channel-planar layout of the per-object blocks: rebuild planar views of the repeated batch
        def planar8(src, c):
            base = src.permute(0, 2, 1).contiguous().repeat(8, 1, 1)      # (8B, C, P) contiguous
            return base.permute(0, 2, 1)                                   # (8B, P, C) strides (C*P, 1, P)
        bx2d, bistd, bx3d = planar8(x2d, 2), planar8(istd, 2), planar8(x3d, 3)
        lb = PnPLaunch(bx2d, bistd, bx3d, K, ur, vr, z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=thr.repeat(8),
                       inlier_opt_only=True, flags=(args.waves << 8))
        for _ in range(3):
            lb.run()
        torch.cuda.synchronize()
        nb = max(4, args.steps // 8)
        t1 = time.perf_counter()
        for _ in range(nb):
            lb.run()
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        extra['single_launch_8192_objects'] = {'value': 8 * B_PER_GPU * nb / el, 'unit': 'solves/s', 'ms_per_launch': el / nb * 1e3}

    if rank == 0:
        total = B_PER_GPU * world * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        achieved = BYTES_PER_SOLVE * B_PER_GPU / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')   # HBM bytes per launch from the PMC passes (see profiles/README.md)
        if os.path.exists(tfile) and not stress:
            try:
                traffic = json.load(open(tfile)).get('hbm_bytes_per_launch')
            except Exception:  # noqa: BLE001
                traffic = None
        valu = None
        sfile = os.path.join(ROOT, 'profiles', 'r01_summary.json')       # PMC instruction counts of the same command
        if os.path.exists(sfile) and not stress:
            try:
                cnt = json.load(open(sfile))['counters']['SQ_INSTS_VALU']['mean']
                # every VALU wave-instruction occupies its SIMD for >= 2 (fp32) .. 4 (fp64) cycles; 1024 SIMDs at 2.4 GHz
                t_min = cnt * 4.0 / (1024 * 2.4e9)
                valu = {'valu_insts_per_launch': cnt, 'min_issue_time_us_at_4_cycles': t_min * 1e6,
                        'frac_of_kernel_time': t_min / (kernel_ms * 1e-3)}
            except Exception:  # noqa: BLE001
                valu = None
        line = {
            'metric': 'PnP solves/sec (1024 proposals, 28x28 corr.)' if not stress else 'PnP solves/sec (stress: 8192 proposals/GPU, 56x56 corr., fp16 storage)', 'value': total / elapsed, 'unit': 'solves/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': ('BASELINE config 2: 1024 synthetic proposals x 28x28 2D-3D correspondences with per-point '
                                    'istd, fp32 storage, channel-planar (NCHW-view) layout, per GPU') if not stress else
                                   ('BASELINE config 5 shard: 8192 synthetic proposals x 56x56 correspondences, fp16 storage, '
                                    'channel-planar layout, per GPU (1024 distinct objects tiled 8x)'),
                       'objects_per_gpu': B_PER_GPU, 'points_per_object': P, 'seed': SEED,
                       'stages': 'istd mask + K0 consensus initialiser (32 hyp.) + LM (Ceres-1.14 semantics, fp64) + covariance',
                       'parallelism': f'objects sharded x{world}' + (', 1 RCCL all-gather of 88 B/object per step' + (' on a side stream, overlapped with the next step' if rccl is not None else '') if world > 1 else '')},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'kernel': 'pnp_uncert_kernel', 'kernel_ms_avg': kernel_ms, 'kernel_ms_min': float(k_ms.min()),
                         'kernel_ms_median': float(np.median(k_ms)), 'kernel_ms_p95': float(np.percentile(k_ms, 95)),
                         'algorithmic_bytes_per_launch': BYTES_PER_SOLVE * B_PER_GPU, 'valu_issue': valu,
                         'note': 'formally HBM-bound (read-once streaming); in practice VALU/latency-bound: the tile is LDS-resident '
                                 'across all LM iterations (DESIGN.md)'},
            'valid_fraction': valid_frac,
            'secondary_throughput': extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            one, allc = cpu_baseline([np.asarray(a) for a in np_inputs], args.cpu_seconds)
            line['cpu_baseline'] = one
            line['cpu_baseline_all_cores'] = allc
            line['speedup_vs_cpu_1thread'] = line['value'] / one['value']
            line['speedup_vs_cpu_all_cores'] = line['value'] / allc['value']
        print(json.dumps(line))
    if rccl is not None:
        rccl.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

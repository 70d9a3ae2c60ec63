"""N>1 path on CPU: world_size-2 gloo run of the sharding + single packed all-gather
(monorun_amd/parallel.py).  The per-shard solve is a stand-in (the CPU oracle — tests may use it;
the product's solve needs an MI355X), what is under test is shard bounds, packing and the exchange."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_objects, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from monorun_amd import synthetic as syn
    from monorun_amd.parallel import sharded_pnp, shard_bounds
    from oracle import oracle as orc
    b = syn.make_batch(B=n_objects, seed=321)             # every rank builds the same global batch
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=False)

    def solve(lo, hi, packed):
        ret, yaw, t, cov, tr, mask = orc.u2d_pnp(x2d[lo:hi], istd[lo:hi], x3d[lo:hi], K, ur, vr, 0.5, 0.6, thr[lo:hi], True)
        n = hi - lo
        packed.pose[:n] = torch.from_numpy(np.concatenate([yaw, t], 1))
        packed.cov[:n] = torch.from_numpy(cov)
        packed.tr[:n] = torch.from_numpy(tr[:, 0])
        packed.valid[:n] = torch.from_numpy(ret.astype(np.uint8))
    out = sharded_pnp(solve, n_objects, torch.device('cpu'))
    lo, hi, per = shard_bounds(n_objects, rank, world)
    q.put((rank, lo, hi, per, {k: v.numpy() for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_objects', [10, 7])
def test_two_rank_sharded_solve_equals_single_process(n_objects):
    from monorun_amd import synthetic as syn
    from oracle import oracle as orc
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_objects, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b = syn.make_batch(B=n_objects, seed=321)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=False)
    ret, yaw, t, cov, tr, mask = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True)
    res.sort(key=lambda r: r[0])
    assert [(r[1], r[2]) for r in res] == [(0, (n_objects + 1) // 2), ((n_objects + 1) // 2, n_objects)]
    for rank, lo, hi, per, out in res:                     # every rank ends up with the full, ordered result
        assert out['pose'].shape == (n_objects, 4) and out['cov'].shape == (n_objects, 4, 4)
        assert np.array_equal(out['pose'], np.concatenate([yaw, t], 1))
        assert np.array_equal(out['cov'], cov) and np.array_equal(out['tr'], tr[:, 0]) and np.array_equal(out['valid'], ret)


def test_shard_bounds_cover_everything_once():
    from monorun_amd.parallel import shard_bounds, PackedResults, ROW_BYTES
    for n in (0, 1, 7, 8, 1024, 65536):
        for w in (1, 2, 4, 8):
            seen = []
            for r in range(w):
                lo, hi, per = shard_bounds(n, r, w)
                assert 0 <= hi - lo <= per and per == (n + w - 1) // w
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    p = PackedResults(5, torch.device('cpu'))
    assert p.buf.numel() == 5 * ROW_BYTES and p.pose.shape == (5, 4) and p.cov.shape == (5, 4, 4)
    p.pose[:] = 1.5; p.cov[:] = 2.5; p.tr[:] = 3.5; p.valid[:] = 1
    u = PackedResults.unpack(p.buf.view(1, -1), 5)
    assert (u['pose'] == 1.5).all() and (u['cov'] == 2.5).all() and (u['tr'] == 3.5).all() and u['valid'].all()


@pytest.mark.gpu
def test_direct_rccl_all_gather_world_size_1():
    """parallel.RcclAllGather (private RCCL communicator, ncclAllGather on a side stream) inside a 1-rank nccl job — the only
    world size a 1-GPU box offers: bootstrap through torch.distributed, stream ordering, byte-exact result."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, ROOT_DIR)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29571')
        torch.cuda.set_device(0); dev = torch.device('cuda', 0)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        from monorun_amd.parallel import RcclAllGather, PackedResults
        ag = RcclAllGather(dev)
        pk = PackedResults(100, dev)
        out = torch.zeros(pk.buf.numel(), dtype=torch.uint8, device=dev)
        for it in range(3):
            pk.pose.copy_(torch.arange(400, device=dev, dtype=torch.float32).view(100, 4) + it)     # produced on the current stream
            pk.valid.fill_(it % 2)
            done = ag.gather(pk.buf, out)
            torch.cuda.current_stream().wait_event(done)
            got = PackedResults.unpack(out.view(1, -1), 100)
            assert torch.equal(got['pose'], pk.pose) and torch.equal(got['valid'], pk.valid.bool()), it
        ag.close(); dist.destroy_process_group(); print('RCCL_OK')
    ''').replace('ROOT_DIR', repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert 'RCCL_OK' in r.stdout, r.stderr[-2000:]

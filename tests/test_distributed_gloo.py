"""N>1 path on CPU: gloo runs (world sizes 2, 4, 8; even, uneven and empty shards) of the sharding + single packed
all-gather (monorun_amd/parallel.py), and of the rank launcher (monorun_amd/launch.py).  On CPU the per-shard solve is a
stand-in (the CPU oracle — tests may use it; the product's solve needs an MI355X): what is under test is shard bounds,
packing, the exchange and the launcher.  The `-m gpu` tests at the bottom run the same `sharded_pnp` with the HIP solve
inside an nccl (RCCL) job and bench.py's self-launch."""
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


@pytest.mark.parametrize('world,n_objects', [(2, 10), (2, 7), (4, 10), (4, 3), (8, 13), (8, 5)])
def test_sharded_solve_equals_single_process(world, n_objects):
    """even shards, uneven shards (last rank short) and ranks whose shard is empty (4 ranks x 3 objects, 8 x 5)."""
    from monorun_amd import synthetic as syn
    from monorun_amd.parallel import shard_bounds
    from oracle import oracle as orc
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_objects, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    b = syn.make_batch(B=n_objects, seed=321)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(b, planar=False)
    ret, yaw, t, cov, tr, mask = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True)
    res.sort(key=lambda r: r[0])
    assert [(r[1], r[2]) for r in res] == [shard_bounds(n_objects, r, world)[:2] for r in range(world)]
    assert sum(r[2] - r[1] for r in res) == n_objects
    for rank, lo, hi, per, out in res:                     # every rank ends up with the full, ordered result
        assert out['pose'].shape == (n_objects, 4) and out['cov'].shape == (n_objects, 4, 4)
        assert np.array_equal(out['pose'], np.concatenate([yaw, t], 1))
        assert np.array_equal(out['cov'], cov) and np.array_equal(out['tr'], tr[:, 0]) and np.array_equal(out['valid'], ret)


def _kitti_val_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('kitti_val', os.path.join(ROOT, 'tools', 'kitti_val.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _oracle_pose_stage(ob, lo, hi):
    """Stand-in for the product's pose stage on CPU ranks: the numpy / C restatement of decode chain + PnP (tests may use the oracle)."""
    from oracle import oracle as orc
    lab = ob['labels'][lo:hi]
    noc, ls, _ = orc.slice_pred(ob['all_pred'][lo:hi], lab, ob['flip'][lo:hi])
    dims, dims_var = orc.dim_decode(ob['dim'][lo:hi], None, lab)
    c3d, c3v = orc.noc_decode(noc, dims, dims_var)
    ls_px = orc.decode_logstd(ls, c3v, exp=orc.spec_expf, log=orc.spec_logf)
    x2d, istd, x3d, ur, vr, thr = orc.pose_head_prep(orc.roi_grid(ob['rois'][lo:hi]), ls_px, c3d, tuple(ob['hw'][lo]), exp=orc.spec_expf)
    ret, yaw, t, cov, _, _ = orc.u2d_pnp(x2d, istd, x3d, np.ascontiguousarray(ob['K'][lo:hi]), ur, vr, 0.5, 0.6, thr, True, num_threads=1)
    return dict(pose=torch.from_numpy(np.concatenate([yaw, t], 1)), cov=torch.from_numpy(cov), valid=torch.from_numpy(ret.astype(np.uint8)),
                dims=torch.from_numpy(dims.astype(np.float32)))


def _oracle_evaluate(results, infos, classes, filenames=None, result_dir=None):
    """Stand-in for monorun_amd.evaluation.evaluate (a HIP evaluator) on CPU ranks: same formatting and result files, AP from the CPU
    restatement of the KITTI protocol."""
    from monorun_amd import evaluation as ev
    from oracle import kitti_eval as ke
    dts = ev.format_results(results, infos, classes)
    os.makedirs(result_dir, exist_ok=True)
    ev.write_result_files(dts, filenames, os.path.join(result_dir, 'data'))
    ap = ke.kitti_ap([ev.format_gt_anno(i, classes) for i in infos], dts, classes, 'R40')
    return ap, 'Car AP (CPU restatement) ' + repr({k: np.round(np.asarray(v, float), 6).tolist() for k, v in ap.items()}), dts


def _harness_worker(rank, world, port, paths, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    kv = _kitti_val_module()
    a = kv.parse_args(['--labels', paths['labels'], '--calib', paths['calib'], '--ids', paths['ids'], '--dumps', paths['dumps'],
                       '--images-per-batch', '3', '--out', os.path.join(os.path.dirname(paths['ids']), f'out_w{world}')])
    ap, text = kv.run(a, pose_fn=_oracle_pose_stage, backend='gloo', dev=torch.device('cpu'), evaluate_fn=_oracle_evaluate)
    q.put((rank, text))


def test_kitti_harness_world_2_equals_one_rank(tmp_path):
    """tools/kitti_val.py's world > 1 branch (BASELINE config 4's shape): the objects of every batch of images split into contiguous
    shards over the ranks, ONE packed all-gather (pose, covariance, validity, dimensions: 100-byte rows) per batch, evaluation on
    rank 0 — run under gloo with two CPU ranks (uneven shards: 15 objects per batch, the last batch 5) and the CPU restatement as
    the pose stage; the KITTI result text equals the one-rank run's."""
    kv = _kitti_val_module()
    paths = kv.write_synthetic_split(str(tmp_path / 'split'), 10, objs_per_img=5)
    ctx = mp.get_context('spawn')
    texts = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_harness_worker, args=(r, world, port, paths, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=600) for _ in range(world))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert all(res[r] is None for r in range(1, world))
        texts[world] = res[0]
    assert texts[1] is not None and 'Car AP' in texts[1] and texts[2] == texts[1]
    assert sorted(os.listdir(tmp_path / 'split' / 'out_w2' / 'data')) == sorted(os.listdir(tmp_path / 'split' / 'out_w1' / 'data'))
    for f in os.listdir(tmp_path / 'split' / 'out_w1' / 'data'):
        assert (tmp_path / 'split' / 'out_w1' / 'data' / f).read_text() == (tmp_path / 'split' / 'out_w2' / 'data' / f).read_text()


def _agree_worker(rank, world, port, broken_rank, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if rank == broken_rank:
        os.environ['MR_RCCL_LIBRARY'] = '/nonexistent/librccl.so'          # ONLY this rank cannot load the library
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from monorun_amd.parallel import agreed_rccl_all_gather
    ag, why = agreed_rccl_all_gather(torch.device('cpu'))
    # the job's next collective still lines up on every rank (nobody is stuck inside a broadcast the other never joined)
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)
    q.put((rank, ag is None, why, int(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('broken_rank', [0, 1])
def test_rccl_set_up_failure_on_one_rank_only_is_agreed(broken_rank):
    """`agreed_rccl_all_gather` must deliver "every rank or none" when only SOME ranks fail.  World size 2 over gloo, one
    rank with MR_RCCL_LIBRARY pointing nowhere: both ranks return (None, reason) — the healthy one because the job agreed, not because it
    failed itself — and the job's following collective completes on both (before the fix the healthy rank sat in the id broadcast while the
    broken one had already moved on to the all-reduce: mismatched collectives)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, broken_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), 'a rank kept a private communicator the other rank does not have'
    assert all(r[2] for r in res) and 'nonexistent' in res[broken_rank][2]
    assert all(r[3] == 3 for r in res)


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
    px = PackedResults(5, torch.device('cpu'), extra_f32=3)            # extra float columns ride in the same row
    assert px.buf.numel() == 5 * (ROW_BYTES + 12) and px.extra.shape == (5, 3)
    px.pose[:] = 1.5; px.extra[:] = 7.25; px.valid[:] = 1
    ux = PackedResults.unpack(torch.cat([px.buf, px.buf]).view(2, -1), 8, extra_f32=3)
    assert ux['extra'].shape == (8, 3) and (ux['extra'] == 7.25).all() and (ux['pose'] == 1.5).all() and ux['valid'].all()
    p = PackedResults(5, torch.device('cpu'))
    assert p.buf.numel() == 5 * ROW_BYTES and p.pose.shape == (5, 4) and p.cov.shape == (5, 4, 4)
    p.pose[:] = 1.5; p.cov[:] = 2.5; p.tr[:] = 3.5; p.valid[:] = 1
    u = PackedResults.unpack(p.buf.view(1, -1), 5)
    assert (u['pose'] == 1.5).all() and (u['cov'] == 2.5).all() and (u['tr'] == 3.5).all() and u['valid'].all()


def test_launcher_starts_n_ranks_and_refuses_without_devices(tmp_path):
    """monorun_amd.launch: `spawn_ranks` starts N ranks through torch.distributed.run on 127.0.0.1 (each sees RANK /
    WORLD_SIZE / LOCAL_RANK and can rendezvous), and refuses with a clear message when N exceeds the visible devices."""
    import subprocess
    from monorun_amd import launch
    script = tmp_path / 'ranks.py'
    script.write_text(
        'import os, sys, torch, torch.distributed as dist\n'
        'dist.init_process_group("gloo")\n'
        't = torch.tensor([float(dist.get_rank() + 1)])\n'
        'dist.all_reduce(t)\n'
        'open(os.path.join(sys.argv[1], "rank%d" % dist.get_rank()), "w").write("%d %d %s %g" % (dist.get_world_size(), int(os.environ["LOCAL_RANK"]), os.environ["MASTER_ADDR"], t.item()))\n'
        'dist.destroy_process_group()\n')
    assert not launch.in_distributed_job()
    rc = launch.spawn_ranks(3, str(script), [str(tmp_path)], need_devices=False)
    assert rc == 0
    for r in range(3):
        assert (tmp_path / f'rank{r}').read_text() == f'3 {r} 127.0.0.1 6'
    cmd = launch.launch_command(8, str(script), ['--x'], port=1234)
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=8' in cmd and cmd[-1] == '--x'
    if launch.visible_devices() < 8:
        with pytest.raises(SystemExit) as e:
            launch.spawn_ranks(8, str(script), [str(tmp_path)])
        assert 'needs 8 visible MI355X devices' in str(e.value)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and 'needs 8 visible MI355X devices' in r.stderr


@pytest.mark.gpu
def test_sharded_pnp_with_the_hip_solve_in_an_nccl_job():
    """parallel.sharded_pnp with the PRODUCT's per-shard solve (the HIP kernel writing straight into PackedResults) inside
    an nccl (= RCCL) torch.distributed job — world size 1 is all a 1-GPU box offers — compared with a plain launch."""
    import subprocess, textwrap
    code = textwrap.dedent('''
        import os, sys, numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, ROOT_DIR)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29573')
        torch.cuda.set_device(0); dev = torch.device('cuda', 0)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        from monorun_amd import synthetic as syn, PnPLaunch
        from monorun_amd.parallel import sharded_pnp
        b = syn.make_batch(B=37, seed=99)
        x2d, istd, x3d, K, ur, vr, thr = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in syn.pnp_boundary(b, planar=False)]
        def solve(lo, hi, packed):
            PnPLaunch(x2d[lo:hi], istd[lo:hi], x3d[lo:hi], K, ur, vr, epnp_ransac_thres=thr[lo:hi], out=packed).run()
        out = sharded_pnp(solve, 37, dev)
        ref = PnPLaunch(x2d, istd, x3d, K, ur, vr, epnp_ransac_thres=thr); ref.run(); torch.cuda.synchronize()
        assert out['pose'].shape == (37, 4) and torch.equal(out['pose'], ref.pose) and torch.equal(out['cov'], ref.cov)
        assert torch.equal(out['valid'], ref.valid.bool()) and torch.equal(out['tr'], ref.tr) and int(out['valid'].sum()) >= 35
        dist.destroy_process_group(); print('SHARDED_HIP_OK')
    ''').replace('ROOT_DIR', repr(ROOT))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert 'SHARDED_HIP_OK' in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` starts its own two ranks.  On a 1-GPU box RCCL refuses two ranks on one device, so the
    test mode MR_BENCH_OVERSUBSCRIBE=1 shares the GPU and exchanges through gloo — launcher, rank environment, per-rank
    batches, max-over-ranks timing and the JSON line (n_gpus, comm block) are the real code path; without the override the
    script refuses with a device-count message.  With >= 2 devices the RCCL path itself runs."""
    import json, subprocess, torch
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    if ndev < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1'],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and 'needs 2 visible MI355X devices' in r.stderr
        env['MR_BENCH_OVERSUBSCRIBE'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--batches', '2',
                        '--no-cpu-baseline', '--no-secondary'], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1000:], r.stderr[-3000:])
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['comm']['nranks'] == 2 and line['comm']['bytes_per_rank'] == 88 * 1024
    assert line['comm']['gathered_rows_verified'] is True and line['value'] > 0 and line['scaling'] == 'weak'
    assert ('rccl' in line['comm']['backend']) == (ndev >= 2)


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
        assert ag.nranks() == 1
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


@pytest.mark.gpu
def test_private_rccl_communicator_through_the_rank_launcher():
    """the private communicator bootstrapped inside a rank that monorun_amd.launch.spawn_ranks started (the
    environment as shipped: HSA_ENABLE_IPC_MODE_LEGACY=0, rendezvous on 127.0.0.1, torch.distributed.run) — world size 1 is what a
    1-GPU box offers; ncclCommCount must agree with the job, and a gather must return this rank's bytes."""
    import tempfile, textwrap
    from monorun_amd import launch
    d = tempfile.mkdtemp(prefix='mr_rccl_')
    script = os.path.join(d, 'rank.py')
    open(script, 'w').write(textwrap.dedent(f'''
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, {ROOT!r})
        rank, world, lr = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
        assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0' and os.environ['MASTER_ADDR'] == '127.0.0.1'
        torch.cuda.set_device(lr); dev = torch.device('cuda', lr)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        from monorun_amd.parallel import agreed_rccl_all_gather, PackedResults
        ag, why = agreed_rccl_all_gather(dev)
        assert ag is not None and why is None, why
        assert ag.nranks() == world == 1
        pk = PackedResults(64, dev); pk.pose.copy_(torch.arange(256, device=dev, dtype=torch.float32).view(64, 4)); pk.valid.fill_(1)
        out = torch.zeros(world * pk.buf.numel(), dtype=torch.uint8, device=dev)
        torch.cuda.current_stream().wait_event(ag.gather(pk.buf, out)); torch.cuda.synchronize()
        assert torch.equal(out, pk.buf)
        ag.close(); dist.destroy_process_group()
        open({os.path.join(d, "ok")!r}, 'w').write('RCCL_LAUNCHED_OK')
    '''))
    assert launch.spawn_ranks(1, script, []) == 0
    assert open(os.path.join(d, 'ok')).read() == 'RCCL_LAUNCHED_OK'


@pytest.mark.gpu
def test_bench_rccl_path_with_launch_sets_and_its_fallback():
    """bench.py inside an RCCL job (MR_BENCH_FORCE_DIST=1: the only way to run that path on a 1-GPU box): the default line — the
    reference flow in launch sets of five calls — exchanges every step's packed rows over the private communicator; when that
    communicator cannot be built (MR_RCCL_LIBRARY points nowhere: VERDICT r4 item 7b) the job falls back to torch.distributed's
    all-gather, says so in comm.backend, and still verifies the gathered rows."""
    import json, subprocess
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '14', '--warmup', '4', '--batches', '3', '--no-cpu-baseline', '--no-secondary']
    for broken in (False, True):
        env = dict(os.environ, MR_BENCH_FORCE_DIST='1', MASTER_PORT=str(29600 + int(broken)))
        if broken:
            env['MR_RCCL_LIBRARY'] = '/nonexistent/librccl.so'
        r = subprocess.run(base, capture_output=True, text=True, timeout=900, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert r.returncode == 0 and len(lines) == 1, (r.stdout[-500:], r.stderr[-3000:])
        d = json.loads(lines[0])
        c = d['comm']
        assert d['config']['flow'].startswith('reference') and d['config']['calls_per_launch_set'] == 5 and d['outputs_verified'] is True
        assert c['nranks'] == 1 and c['bytes_per_rank'] == 88 * 1024 and c['gathered_rows_verified'] is True
        if broken:
            assert 'FALLBACK' in c['backend'] and 'torch.distributed' in c['backend'] and 'direct RCCL path unavailable' in r.stderr
        else:
            assert c['backend'].startswith('rccl (private communicator') and c['us_per_collective'] > 0

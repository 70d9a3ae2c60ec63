"""Single-node rank launcher: one process per GPU, rendezvous on 127.0.0.1.

The reference starts its own workers (``/root/reference/train.py:67-74`` spawns one process per GPU through
``torch.multiprocessing``); here a script that is asked for N GPUs and finds itself outside a
``torch.distributed`` job re-launches itself as N ranks through ``torch.distributed.run`` (the same module the
driver uses), so ``python bench.py --gpus 8`` and ``python tools/kitti_val.py --gpus 8`` work unattended.
"""
import os
import socket
import subprocess
import sys


def in_distributed_job():
    """True inside a rank started by torch.distributed.run / torchrun (RANK and WORLD_SIZE are exported)."""
    return 'RANK' in os.environ and 'WORLD_SIZE' in os.environ


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_devices():
    """Number of HIP devices this process can see, asked of the C ABI (no torch.cuda initialisation in the parent)."""
    from . import _lib
    return int(_lib.load().mr_pnp_device_count())


def launch_command(nproc, script, argv, port=None):
    """The exact command line ``spawn_ranks`` executes (also what a user would type by hand)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(nproc)}',
            '--master-addr', '127.0.0.1', '--master-port', str(port if port is not None else free_port()),
            os.path.abspath(script)] + list(argv)


def spawn_ranks(nproc, script, argv, need_devices=True, env=None):
    """Run `script argv` as `nproc` ranks on this node and return the launcher's exit code.

    need_devices: refuse (SystemExit with a clear message) when fewer than `nproc` HIP devices are visible —
    RCCL cannot place two ranks on one device.  Rank output is passed through unchanged (rank 0 prints the
    JSON line)."""
    nproc = int(nproc)
    if need_devices:
        have = visible_devices()
        if have < nproc:
            raise SystemExit(f'{os.path.basename(script)}: --gpus {nproc} needs {nproc} visible MI355X devices, this box has {have} '
                             f'(one rank per GPU over RCCL; check HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES)')
    e = dict(os.environ)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    e.setdefault('OMP_NUM_THREADS', '1')
    if env:
        e.update(env)
    return subprocess.call(launch_command(nproc, script, argv), env=e)

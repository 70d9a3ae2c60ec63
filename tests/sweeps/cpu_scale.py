import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from monorun_amd import synthetic as syn
from oracle import oracle as orc
b = syn.make_batch(B=1024, seed=1234)
x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
print('affinity', len(os.sched_getaffinity(0)), 'OMP env', os.environ.get('OMP_NUM_THREADS'))
try: print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print('no cgroup cpu.max', e)
for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 1.5:
        orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, num_threads=n); reps += 1
    el = time.perf_counter() - t0
    print(n, 'threads:', 1024 * reps / el, 'solves/s')

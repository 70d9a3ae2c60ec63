"""µs per call of the reference's three host-buffer entry points (ext.h) as this library serves them (development aid)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from monorun_amd import _lib, synthetic as syn
lib = _lib.load()
dp = ctypes.POINTER(ctypes.c_double)
b = syn.make_batch(B=8, seed=5)
x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
for pn in (64, 600, 784):
    p2, p3, w = [np.ascontiguousarray(a[0][:pn], np.float64) for a in (x2d, x3d, istd)]
    Kd = np.ascontiguousarray(K[0], np.float64).reshape(9); init = np.array([b['gt_yaw'][0] + 0.05, *(b['gt_t'][0] + 0.2)])
    clips = np.array([0.5, -200, 1442, -200, 575.]); val = np.zeros(1, np.int32); pose = np.zeros(4); cov = np.eye(4); tr = np.zeros(1)
    args = [a.ctypes.data_as(dp) for a in (p2, p3, w, Kd, init)] + [val.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), pose.ctypes.data_as(dp), cov.ctypes.data_as(dp), tr.ctypes.data_as(dp), pn, clips.ctypes.data_as(dp)]
    for _ in range(20): lib.pnp_uncert(*args)
    t0 = time.perf_counter()
    for _ in range(500): lib.pnp_uncert(*args)
    print(f'pnp_uncert (ext.h:1-13), pn = {pn}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call (valid {val[0]})')
import test_noc_variants as tn
for full_cov in (False, True):
    q = tn._problem(full_cov, seed=1, n=600)
    name = 'pnp_noc_cov_uncert' if full_cov else 'pnp_noc_uncert'
    for _ in range(10): tn._call(lib, name, q)
    t0 = time.perf_counter()
    for _ in range(200): tn._call(lib, name, q)
    print(f'{name} (ext.h:15-43), pn = 600: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call (incl. the Python marshalling of the test helper)')

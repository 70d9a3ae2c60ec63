import sys, os, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from monorun_amd import synthetic as syn, _lib
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_device
dev = torch.device('cuda:0'); lib = _lib.load(); lib.mr_pnp_debug_set_stamps.argtypes = [ctypes.c_void_p]
import itertools
for B, W in itertools.product((1, 1024), (4, 2)):
    b = syn.make_batch(B=B, seed=1234)
    def dv(a):
        t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
    x = [dv(a) for a in syn.pnp_boundary(b, planar=True)]
    st = torch.zeros(B, 24, dtype=torch.int64, device=dev)
    for it in range(3):
        lib.mr_pnp_debug_set_stamps(st.data_ptr())
        pnp_uncert_device(x[0], x[1], x[2], x[3], x[4], x[5], 0.5, 0.6, x[6], True, flags=W << 8)
        torch.cuda.synchronize()
    lib.mr_pnp_debug_set_stamps(None)
    s = st.cpu().numpy().astype(np.float64)
    print('B', B, 'waves per object', W, 'kernel start->records in LDS %d ' % np.median(s[:, 1] - s[:, 0]), 'load->sums %d  sums->ballots %d  ballots->list %d  (median cycles)' % (np.median(s[:, 10] - s[:, 1]), np.median(s[:, 11] - s[:, 10]), np.median(s[:, 2] - s[:, 11])))

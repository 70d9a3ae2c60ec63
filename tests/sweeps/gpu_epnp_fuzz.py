"""Adversarial inputs for the EPnP / RANSAC initialiser path (pnp_uncert(..., initialiser='epnp')): the kernels must terminate, never
report a non-finite pose as valid, and agree with the reference's flow restated on the success flags and inlier masks (development
aid; run under `timeout`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from monorun_amd import synthetic as syn
from monorun_amd.ops.least_squares.pnp_uncert import pnp_uncert_epnp_device
from oracle import oracle as orc
from tests import fuzz_cases
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(os.environ.get('SEED', 0)))
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
bad = 0
nobj = 0
nbad_valid = 0
for trial in range(int(os.environ.get('TRIALS', 40))):
    mode = trial % 10
    x2d, istd, x3d, K, ur, vr, thr = fuzz_cases.make_case(mode, rng)
    B, P = x2d.shape[:2]
    with np.errstate(all='ignore'):
        ref = orc.u2d_pnp_epnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, return_init=True, num_threads=0)
    d = [dv(x2d), dv(istd), dv(x3d), dv(K), dv(ur), dv(vr), dv(thr)]
    out = pnp_uncert_epnp_device(d[0], d[1], d[2], d[3], d[4], d[5], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=d[6], inlier_opt_only=True, with_diag=True)
    ini, iv = out[6], out[7]
    torch.cuda.synchronize()
    valid, pose, mask = out[0].cpu().numpy().astype(bool), out[1].cpu().numpy(), out[4].cpu().numpy().astype(bool)
    nobj += B
    if not np.isfinite(pose[valid]).all():
        bad += 1; print('trial', trial, 'mode', mode, 'non-finite pose reported valid')
    init_ok_ref = ref[6][:, 2] != 8
    if not np.array_equal(iv.cpu().numpy().astype(bool), init_ok_ref):
        dd = np.flatnonzero(iv.cpu().numpy().astype(bool) != init_ok_ref)
        print('trial', trial, 'mode', mode, 'B', B, 'P', P, 'initialiser success flag differs for', len(dd), 'objects', dd[:5]); bad += 1
    if not np.array_equal(mask, ref[5]):
        dd = np.flatnonzero((mask != ref[5]).any(1))
        print('trial', trial, 'mode', mode, 'B', B, 'P', P, 'inlier mask differs for', len(dd), 'objects', dd[:5]); bad += 1
    if not np.array_equal(valid, ref[0]):
        dd = np.flatnonzero(valid != ref[0])
        print('trial', trial, 'mode', mode, 'B', B, 'P', P, 'validity differs from the restatement for', len(dd), 'objects', dd[:5]); bad += 1
        nbad_valid += len(dd)
        if os.environ.get('DETAIL'):
            gd = out[5].cpu().numpy()
            for o in dd[:3]:
                print(f'    object {o}: restatement valid {bool(ref[0][o])} iters {ref[6][o, 0]:.0f} why {ref[6][o, 2]:.0f} cost {ref[6][o, 1]:.6g} | GPU valid {bool(valid[o])} iters {gd[o, 0]:.0f} why {gd[o, 2]:.0f} '
                      f'cost {gd[o, 1]:.6g} | pose ref {ref[1][o].tolist() + ref[2][o].tolist()} gpu {pose[o].tolist()} | cov diag ref {np.diag(ref[3][o]).tolist()} gpu {np.diag(out[2][o].cpu().numpy()).tolist()}')
print('EPnP fuzz done:', nobj, 'objects, trials with a problem:', bad, '; objects whose valid flag differs:', nbad_valid)

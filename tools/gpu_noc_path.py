"""The NOC path at B = 1024 for the profiler (development aid; tools/profile_round.sh): K2 (noc_decode_kernel), the PnP kernel fed by
its output, and the fused head->pose kernel, each launched REPS times over 3 distinct resident head outputs (276 MiB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch
from monorun_amd.pose_head import NocDecodeLaunch, PoseFromHeadLaunch, UncertPropPnPOptimizer, _planar_view, _clip_ranges
dev = torch.device('cuda:0')
REPS = int(os.environ.get('REPS', 40))
WHICH = os.environ.get('WHICH', 'k2,pnp,fused').split(',')
head = UncertPropPnPOptimizer().to(dev)
k2s, pnps, fus = [], [], []
for i in range(3):
    b = syn.make_batch(B=1024, hw=28, seed=1234 + 7919 * i)
    all_pred, dim = syn.encode_head_outputs(b, seed=1234 + i)
    rng = np.random.default_rng(1234 + i)
    dim_var = torch.from_numpy((0.01 * rng.random((1024, 3)) + 1e-4).astype(np.float32)).to(dev)
    ap, lab, dm, rois, K = [torch.from_numpy(x).to(dev) for x in (all_pred, b['labels'], dim, b['rois'], b['K'])]
    k2 = NocDecodeLaunch(ap, lab, False, dm, dim_var, rois)
    ur, vr = _clip_ranges((syn.IMG_H, syn.IMG_W), head.allowed_border, dev)
    d = k2.out
    pnps.append(PnPLaunch(_planar_view(d['coords_2d']), _planar_view(d['coords_2d_istd']), _planar_view(d['coords_3d']), K, ur, vr, z_min=0.5,
                          epnp_istd_thres=0.6, epnp_ransac_thres=d['ransac_thr'], inlier_opt_only=True))
    fus.append(PoseFromHeadLaunch(head, ap, lab, False, dm, dim_var, rois, K, (syn.IMG_H, syn.IMG_W)))
    k2s.append(k2)
torch.cuda.synchronize()
for r in range(REPS):
    i = r % 3
    if 'k2' in WHICH: k2s[i].run()
    if 'pnp' in WHICH: pnps[i].run()
    if 'fused' in WHICH: fus[i].run()
torch.cuda.synchronize()
print('ok', REPS)

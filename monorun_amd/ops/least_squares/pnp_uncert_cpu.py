"""numpy-level batch driver — drop-in for ``u2d_pnp_cpu`` of
/root/reference/monorun/ops/least_squares/pnp_uncert_cpu.py:128-209 (same name, arguments and
6-tuple).  Despite the inherited name nothing is solved on the CPU: the arrays are staged to the
MI355X (keeping their strides, which select numpy's summation order for the istd mean), the fused
HIP kernel runs, and the results come back as numpy arrays.
"""
import numpy as np
import torch

from ... import _lib
from .pnp_uncert import DEFAULT_INITIALISER, pnp_uncert_device, pnp_uncert_epnp_device


def _to_dev(a, dev):
    """numpy -> device tensor with the SAME element strides (no re-layout)."""
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float16, np.float64):
        a = a.astype(np.float32)
    if not a.flags.writeable:
        a = a.copy()
    t = torch.from_numpy(a)
    d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev)
    d.copy_(t)
    return d


def u2d_pnp_cpu(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=0.5, epnp_istd_thres=1.0,
                epnp_ransac_thres=None, inlier_opt_only=False, with_pose_cov=True, initialiser=None, epnp_first_round=None):
    """Batched pose solve on numpy arrays (keyword names and defaults are the reference's, pnp_uncert_cpu.py:128-135).

    Inputs, B objects with P correspondences each:
      coords_2d (B,P,2) image points, coords_2d_istd (B,P,2) their inverse standard deviations, coords_3d (B,P,3)
      object-frame points; cam_mats (B|1,3,3); u_range / v_range (B|1,2) clip intervals of the projection; z_min depth
      clamp; epnp_istd_thres the istd-inlier factor; epnp_ransac_thres (B,) consensus thresholds in pixels or None;
      inlier_opt_only: refine on the inlier set only; with_pose_cov: also return the covariance.
      initialiser (not a reference keyword): 'epnp' (the default since round 5) = the reference's — cv2.solvePnPRansac(...,
      iterationsCount=30, flags=SOLVEPNP_EPNP) / cv2.solvePnP without a threshold (pnp_uncert_cpu.py:33-68), restated on the GPU — in
      front of the same LM: the flow this function has in the reference; 'k0' = the fast mode, the fused kernel's own deterministic
      consensus initialiser (one launch).
      epnp_first_round: see ``pnp_uncert``.
    Output 6-tuple (float32 / bool numpy arrays):
      ret_val (B,) success flags, yaw (B,1), t_vec (B,3), pose_cov (B,4,4) = (J^T J)^-1 of [yaw, t] with the solver's
      Jacobian, as Ceres' Covariance reports it (pnp_uncert_cpu.cpp:279-291), or None; tr_radius (B,1) final trust-region
      radius; inlier_mask (B,P).
    """
    bn = coords_2d.shape[0]
    pn = coords_2d.shape[1]
    if bn == 0:
        return (np.zeros((0, ), dtype=bool), np.zeros((0, 1), dtype=np.float32), np.zeros((0, 3), dtype=np.float32),
                np.zeros((0, 4, 4), dtype=np.float32), np.zeros((0, 1), dtype=np.float32), np.zeros((0, pn), dtype=bool))
    assert coords_2d_istd.shape[1] == coords_3d.shape[1] == pn >= 4
    if not torch.cuda.is_available():
        raise RuntimeError('monorun_amd.ops.u2d_pnp_cpu needs an MI355X (HIP) device; no CPU fallback exists')
    dev = torch.device('cuda', torch.cuda.current_device())
    flags = _lib.MR_COV_CERES if with_pose_cov else _lib.MR_COV_NONE
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    x2d, istd, x3d, cam, ur, vr = _to_dev(coords_2d, dev), _to_dev(coords_2d_istd, dev), _to_dev(coords_3d, dev), t(cam_mats), t(u_range), t(v_range)
    thr = t(epnp_ransac_thres) if epnp_ransac_thres is not None else None
    if initialiser is None:
        initialiser = DEFAULT_INITIALISER
    if initialiser == 'epnp':
        valid, pose, cov, tr, mask = pnp_uncert_epnp_device(x2d, istd, x3d, cam, ur, vr, z_min=z_min, epnp_istd_thres=epnp_istd_thres, epnp_ransac_thres=thr,
                                                            inlier_opt_only=inlier_opt_only, flags=flags, first_round=epnp_first_round)[:5]
    elif initialiser == 'k0':
        valid, pose, cov, tr, mask, _ = pnp_uncert_device(x2d, istd, x3d, cam, ur, vr, z_min=z_min, epnp_istd_thres=epnp_istd_thres,
                                                          epnp_ransac_thres=thr, inlier_opt_only=inlier_opt_only, flags=flags)
    else:
        raise ValueError(f"initialiser must be 'k0' or 'epnp', got {initialiser!r}")
    ret_val = valid.cpu().numpy().astype(bool)
    pose = pose.cpu().numpy()
    if with_pose_cov:
        pose_cov = cov.cpu().numpy()
        pose_cov[~ret_val] = np.eye(4, dtype=np.float32)      # result_cov keeps its identity init on failure (:93,:119-125)
    else:
        pose_cov = None
    return (ret_val, pose[:, :1].copy(), pose[:, 1:].copy(), pose_cov,
            tr.cpu().numpy()[:, None], mask.cpu().numpy().astype(bool))

"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/pnp_oracle.c header).

Python face of the CPU restatement: ctypes bindings to ``libpnp_oracle.so`` plus numpy
restatements of the host-side / elementwise stages of the reference hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
``monorun_amd`` never does.

Reference rows restated here in numpy, paths relative to /root/reference:
  R4  istd inlier mask ........ monorun/ops/least_squares/pnp_uncert_cpu.py:164-168
  R8  pose-head input prep .... monorun/models/roi_heads/bbox_3d_heads/optimizers/uncert_prop_pnp_optimizer.py:73-97
  R9  flip/class channel pick . monorun/models/roi_heads/bbox_3d_heads/dense_decoders/fcn_noc_decoder.py:225-267
  R10 dim / NOC decode ........ monorun/core/bbox_3d/dim_coder/multiclass_norm_dim_coder.py:28-36,
                                monorun/core/bbox_3d/coord_coder/noc_coder.py:50-73
  R11 log-std decode .......... monorun/core/bbox_3d/proj_error_coder/distance_invar_proj_error_coder.py:39-60
  R12 roi_align(coord_2d) ..... monorun/models/roi_heads/monorun_roi_head.py:521-523 (`roi_grid`: analytic interior form;
                                `roi_align_avg`: the published mmcv algorithm restated — mmcv absent -> parity UNPINNED,
                                checked against closed forms in tests/test_roi_align.py)
  R13 cov_correction .......... distance_invar_proj_error_coder.py:62-63
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = ctypes.POINTER(ctypes.c_double)
c_fp = ctypes.POINTER(ctypes.c_float)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    so = os.path.join(_HERE, 'libpnp_oracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('pnp_oracle.c', 'epnp.inc', 'jet.inc', 'Makefile')]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libpnp_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_max_threads.restype = ctypes.c_int
        _LIB.orc_k0_init.restype = ctypes.c_int
        _LIB.orc_pose_cov.restype = ctypes.c_int
        _LIB.orc_epnp_ransac.restype = ctypes.c_int
        _LIB.orc_epnp_ransac_trace.restype = ctypes.c_int
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


# ----------------------------------------------------------------------------- C restatement ---
def pnp_uncert(pts2d, pts3d, wgt2d, K, init_pose, clips, with_cov=False):
    """The reference's C entry point (src/ext.h:1-13): returns dict(val, pose, cov, tr, diag)."""
    pts2d, pts3d, wgt2d, K, init_pose, clips = map(_d, (pts2d, pts3d, wgt2d, K, init_pose, clips))
    pn = pts2d.shape[0]
    val = np.zeros(1, np.int32)
    pose = np.zeros(4)
    cov = np.eye(4) if with_cov else None
    tr = np.zeros(1)
    diag = np.zeros(6)
    lib().orc_pnp_uncert_diag(_p(pts2d, c_dp), _p(pts3d, c_dp), _p(wgt2d, c_dp), _p(K, c_dp),
                              _p(init_pose, c_dp), _p(val, c_ip), _p(pose, c_dp), _p(cov, c_dp),
                              _p(tr, c_dp), ctypes.c_int(pn), _p(clips, c_dp), _p(diag, c_dp))
    return dict(val=int(val[0]), pose=pose, cov=cov, tr=float(tr[0]),
                iters=int(diag[0]), why=int(diag[1]), termination=int(diag[2]),
                initial_cost=diag[3], final_cost=diag[4], n_success=int(diag[5]))


TRACE_FIELDS = ('iteration', 'cost', 'candidate_cost', 'model_cost_change', 'relative_decrease', 'radius_after', 'step_norm', 'outcome')
# outcome: 1 accepted, 0 rejected, -1 invalid step (radius halved), -2 fifth invalid step (failure), 2 / 3 parameter / function tolerance exit


def pnp_uncert_opt(pts2d, pts3d, wgt2d, K, init_pose, clips, qr=False, max_iter=50, trace=False):
    """`pnp_uncert` with explicit LM options: qr=True solves every trust-region step the way Ceres' DENSE_QR does (Householder QR of
    [J S; D]) instead of through the normal equations; max_iter = Ceres' max_num_iterations; trace=True also returns the per-pass
    record (rows of TRACE_FIELDS)."""
    pts2d, pts3d, wgt2d, K, init_pose, clips = map(_d, (pts2d, pts3d, wgt2d, K, init_pose, clips))
    pn = pts2d.shape[0]
    val, pose, tr, diag = np.zeros(1, np.int32), np.zeros(4), np.zeros(1), np.zeros(6)
    cap = 64 if trace else 0
    tbuf = np.full((max(cap, 1), len(TRACE_FIELDS)), np.nan)
    f = lib().orc_pnp_uncert_opt
    f.restype = ctypes.c_int
    n = f(_p(pts2d, c_dp), _p(pts3d, c_dp), _p(wgt2d, c_dp), _p(K, c_dp), _p(init_pose, c_dp), _p(val, c_ip), _p(pose, c_dp),
          _p(tr, c_dp), ctypes.c_int(pn), _p(clips, c_dp), _p(diag, c_dp), ctypes.c_int(bool(qr)), ctypes.c_int(max_iter),
          _p(tbuf, c_dp) if trace else None, ctypes.c_int(cap))
    out = dict(val=int(val[0]), pose=pose, tr=float(tr[0]), iters=int(diag[0]), why=int(diag[1]), termination=int(diag[2]),
               initial_cost=diag[3], final_cost=diag[4], n_success=int(diag[5]))
    if trace:
        out['trace'] = tbuf[:n].copy()
    return out


def set_lm_options(qr=False, max_iter=50):
    """Process-wide LM options of the batch driver (u2d_pnp): step solver and max_num_iterations.  Reset with set_lm_options()."""
    lib().orc_set_lm_options(ctypes.c_int(bool(qr)), ctypes.c_int(max_iter))


def residual_jacobian(K, clips, pose, pts2d, pts3d, wgt2d):
    """R1: Ceres-semantics residuals (pn,2) and Jacobian (pn,2,4)."""
    pts2d, pts3d, wgt2d, K, pose, clips = map(_d, (pts2d, pts3d, wgt2d, K, pose, clips))
    pn = pts2d.shape[0]
    res = np.zeros((pn, 2))
    jac = np.zeros((pn, 2, 4))
    lib().orc_residual_jacobian(_p(K, c_dp), _p(clips, c_dp), _p(pose, c_dp), _p(pts2d, c_dp),
                                _p(pts3d, c_dp), _p(wgt2d, c_dp), ctypes.c_int(pn), _p(res, c_dp), _p(jac, c_dp))
    return res, jac


def torch_jacobian(K, z_min, u_range, v_range, yaw, t, pts2d, pts3d, istd, inlier=None):
    """R2/R7 for one object: jac (pn,2,4) [yaw,tx,ty,tz], weighted err (pn,2), H = J^T J (4,4)."""
    pts2d, pts3d, istd, K, u_range, v_range, t = map(_d, (pts2d, pts3d, istd, K, u_range, v_range, t))
    pn = pts2d.shape[0]
    jac = np.zeros((pn, 2, 4))
    err = np.zeros((pn, 2))
    H = np.zeros((4, 4))
    inl = np.ascontiguousarray(inlier, np.uint8) if inlier is not None else None
    lib().orc_torch_jacobian(_p(K, c_dp), ctypes.c_double(z_min), _p(u_range, c_dp), _p(v_range, c_dp),
                             ctypes.c_double(float(yaw)), _p(t, c_dp), _p(pts2d, c_dp), _p(pts3d, c_dp),
                             _p(istd, c_dp), _p(inl, c_u8p), ctypes.c_int(pn), _p(jac, c_dp), _p(err, c_dp), _p(H, c_dp))
    return jac, err, H


def exact_hessian(K, z_min, u_range, v_range, yaw, t, pts2d, pts3d, istd, inlier=None):
    """hessian.py:5-64 for one object (closed form of what the reference gets by autograd): h (4,4) fp64."""
    pts2d, pts3d, istd, K, u_range, v_range, t = map(_d, (pts2d, pts3d, istd, K, u_range, v_range, t))
    H = np.zeros((4, 4))
    inl = np.ascontiguousarray(inlier, np.uint8) if inlier is not None else None
    lib().orc_exact_hessian(_p(K, c_dp), ctypes.c_double(z_min), _p(u_range, c_dp), _p(v_range, c_dp), ctypes.c_double(float(yaw)),
                            _p(t, c_dp), _p(pts2d, c_dp), _p(pts3d, c_dp), _p(istd, c_dp), _p(inl, c_u8p), ctypes.c_int(pts2d.shape[0]),
                            _p(H, c_dp))
    return H


def pose_cov_general(H):
    """inverse(h) for a general (possibly indefinite) h: (ok, cov); singular -> (False, identity)."""
    H = _d(H)
    cov = np.zeros((4, 4))
    ok = lib().orc_pose_cov_general(_p(H, c_dp), _p(cov, c_dp))
    return bool(ok), cov


def pose_cov(H):
    H = _d(H)
    cov = np.zeros((4, 4))
    ok = lib().orc_pose_cov(_p(H, c_dp), _p(cov, c_dp))
    return bool(ok), cov


def cov_symeig_rule(valid, cov):
    """The reference's eigenvalue rule (pnp_uncert.py:77-85) per object, restated on cov = h^-1 (reciprocal eigenvalues):
    keep iff lambda_min(cov) > max(1e-6 lambda_max(cov), 0); otherwise valid = False, cov = I.  Returns (valid, cov, eigs (B,2))."""
    valid, cov = np.array(valid, bool), np.array(cov, np.float32)
    lam = np.zeros((len(valid), 2))
    for i in range(len(valid)):
        c = cov[i].astype(np.float64)
        w = np.linalg.eigvalsh(0.5 * (c + c.T)) if np.isfinite(c).all() else np.array([np.nan] * 4)
        lam[i] = (w[0], w[-1])
        if not (np.isfinite(c).all() and w[0] > max(1e-6 * w[-1], 0.0)):
            valid[i] = False
            cov[i] = np.eye(4, dtype=np.float32)
    return valid, cov, lam


def k0_init(x2d, x3d, mask0, K, ransac_thr=None, n_hyp=32):
    """K0 for one object.  Returns dict(ok, init_pose, mask, best_hyp, best_count)."""
    x2d, x3d, K = _f(x2d), _f(x3d), _f(K)
    mask = np.ascontiguousarray(mask0, np.uint8).copy()
    init = np.zeros(4)
    bh = np.zeros(1, np.int32)
    bc = np.zeros(1, np.int32)
    ok = lib().orc_k0_init(_p(x2d, c_fp), _p(x3d, c_fp), _p(mask, c_u8p), ctypes.c_int(x2d.shape[0]), _p(K, c_fp),
                           ctypes.c_int(ransac_thr is not None), ctypes.c_float(0.0 if ransac_thr is None else float(ransac_thr)),
                           ctypes.c_int(n_hyp), _p(init, c_dp), _p(bh, c_ip), _p(bc, c_ip))
    return dict(ok=bool(ok), init_pose=init, mask=mask.astype(bool), best_hyp=int(bh[0]), best_count=int(bc[0]))


def istd_inlier_mask(coords_2d_istd, epnp_istd_thres):
    """R4 (pnp_uncert_cpu.py:164-168) — literally the reference's numpy expression, so the float32
    summation order numpy picks for the array's strides is the reference's own."""
    mean = np.mean(coords_2d_istd, axis=1, keepdims=True)
    return np.min(coords_2d_istd >= epnp_istd_thres * mean, axis=2)


def u2d_pnp(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=0.5,
            epnp_istd_thres=1.0, epnp_ransac_thres=None, inlier_opt_only=False,
            init_pose=None, n_hyp=32, num_threads=1, return_diag=False, init_mode=0, return_init=False, return_pose64=False):
    """R4+R5+R6 for a batch (the numpy-level driver).  Returns the reference's 6-tuple
    (ret_val, yaw, t_vec, pose_cov, tr_radius, inlier_mask) [+ diag] [+ init (B,4) fp64]; pose_cov is the
    torch-semantics inverse(J^T J) that pnp_uncert.py:71-85 computes.
    init_mode 0: K0 (this repo's consensus initialiser, what the HIP kernel runs); 1: the reference's own initialiser
    restated (EPnP inside OpenCV's RANSAC loop, oracle/epnp.inc)."""
    B, P = coords_2d.shape[:2]
    if B == 0:
        out = (np.zeros((0,), bool), np.zeros((0, 1), np.float32), np.zeros((0, 3), np.float32),
               np.zeros((0, 4, 4), np.float32), np.zeros((0, 1), np.float32), np.zeros((0, P), bool))
        out = out + (np.zeros((0, 4), np.float32),) if return_diag else out
        return out + (np.zeros((0, 4)),) if return_init else out
    assert coords_2d_istd.shape[1] == coords_3d.shape[1] == P >= 4
    mask = np.ascontiguousarray(istd_inlier_mask(coords_2d_istd, epnp_istd_thres), np.uint8)
    x2d, istd, x3d = _f(coords_2d), _f(coords_2d_istd), _f(coords_3d)
    K = _f(cam_mats).reshape(-1, 9)
    ur, vr = _f(u_range).reshape(-1, 2), _f(v_range).reshape(-1, 2)
    assert ur.shape[0] == vr.shape[0]
    thr = _f(epnp_ransac_thres) if epnp_ransac_thres is not None else None
    ini = _d(init_pose) if init_pose is not None else None
    valid = np.zeros(B, np.uint8)
    pose = np.zeros((B, 4), np.float32)
    cov = np.zeros((B, 16), np.float32)
    tr = np.zeros(B, np.float32)
    diag = np.zeros((B, 4), np.float32)
    init_out = np.zeros((B, 4))
    pose64 = np.zeros((B, 4))
    lib().orc_u2d_pnp_batch_ex(_p(x2d, c_fp), _p(istd, c_fp), _p(x3d, c_fp), _p(K, c_fp), ctypes.c_int(K.shape[0]),
                               _p(ur, c_fp), _p(vr, c_fp), ctypes.c_int(ur.shape[0]), _p(thr, c_fp), _p(ini, c_dp),
                               ctypes.c_int(B), ctypes.c_int(P), ctypes.c_double(z_min), ctypes.c_int(bool(inlier_opt_only)),
                               ctypes.c_int(n_hyp), ctypes.c_int(init_mode), ctypes.c_int(num_threads), _p(mask, c_u8p), _p(valid, c_u8p),
                               _p(pose, c_fp), _p(cov, c_fp), _p(tr, c_fp), _p(diag, c_fp), _p(init_out, c_dp), _p(pose64, c_dp))
    out = (valid.astype(bool), pose[:, :1].copy(), pose[:, 1:].copy(), cov.reshape(B, 4, 4),
           tr[:, None].copy(), mask.astype(bool))
    out = out + (diag,) if return_diag else out
    out = out + (init_out,) if return_init else out
    return out + (pose64,) if return_pose64 else out


def u2d_pnp_epnp(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=0.5, epnp_istd_thres=1.0,
                 epnp_ransac_thres=None, inlier_opt_only=False, **kw):
    """The reference's flow with its OWN initialiser restated (cv2.solvePnPRansac / solvePnP with SOLVEPNP_EPNP,
    pnp_uncert_cpu.py:35-58) in front of the same LM and covariance: the comparison point for K0."""
    return u2d_pnp(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min, epnp_istd_thres, epnp_ransac_thres,
                   inlier_opt_only, init_mode=1, **kw)


def epnp(obj, img, K):
    """cv2.solvePnP(obj, img, K, 0, flags=SOLVEPNP_EPNP) restated: returns (rvec (3,), tvec (3,), R (3,3))."""
    obj, img, K = _f(obj), _f(img), _f(K).reshape(9)
    rvec, tvec, R = np.zeros(3), np.zeros(3), np.zeros(9)
    lib().orc_epnp(_p(obj, c_fp), _p(img, c_fp), ctypes.c_int(obj.shape[0]), _p(K, c_fp), _p(rvec, c_dp), _p(tvec, c_dp), _p(R, c_dp))
    return rvec, tvec, R.reshape(3, 3)


def epnp_ransac(obj, img, K, thr, max_iters=30):
    """cv2.solvePnPRansac(obj, img, K, 0, reprojectionError=thr, iterationsCount=max_iters, flags=SOLVEPNP_EPNP) restated:
    returns dict(ok, rvec, tvec, mask (n,) bool, iters)."""
    obj, img, K = _f(obj), _f(img), _f(K).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    mask = np.zeros(obj.shape[0], np.uint8)
    it = np.zeros(1, np.int32)
    ok = lib().orc_epnp_ransac(_p(obj, c_fp), _p(img, c_fp), ctypes.c_int(obj.shape[0]), _p(K, c_fp), ctypes.c_float(thr), ctypes.c_int(max_iters),
                               _p(rvec, c_dp), _p(tvec, c_dp), _p(mask, c_u8p), _p(it, c_ip))
    return dict(ok=bool(ok), rvec=rvec, tvec=tvec, mask=mask.astype(bool), iters=int(it[0]))


def epnp_ransac_trace(obj, img, K, thr, max_iters=30):
    """epnp_ransac plus, for every RANSAC iteration that ran, the hypothesis (R after the Rodrigues round trip | t, (max_iters,12)) and
    its inlier count ((max_iters,), -1 = not evaluated because the adaptive iteration count had already been reached)."""
    obj, img, K = _f(obj), _f(img), _f(K).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    mask = np.zeros(obj.shape[0], np.uint8)
    it = np.zeros(1, np.int32)
    hyp, cnt = np.zeros((max_iters, 12)), np.zeros(max_iters, np.int32)
    ok = lib().orc_epnp_ransac_trace(_p(obj, c_fp), _p(img, c_fp), ctypes.c_int(obj.shape[0]), _p(K, c_fp), ctypes.c_float(thr), ctypes.c_int(max_iters),
                                     _p(rvec, c_dp), _p(tvec, c_dp), _p(mask, c_u8p), _p(it, c_ip), _p(hyp, c_dp), _p(cnt, c_ip))
    return dict(ok=bool(ok), rvec=rvec, tvec=tvec, mask=mask.astype(bool), iters=int(it[0]), hyp=hyp, cnt=cnt)


def eig_sym(A):
    A = _d(A); n = A.shape[0]
    w, vt = np.zeros(n), np.zeros((n, n))
    lib().orc_eig_sym(ctypes.c_int(n), _p(A, c_dp), _p(w, c_dp), _p(vt, c_dp))
    return w, vt


def eig12(A):
    """The specification's eigen-solver for the 12 x 12 M^T M (Householder tridiagonalisation + implicit QL, epnp.inc::epnp_eig12):
    eigenvalues descending, eigenvectors in the rows of vt."""
    A = _d(A)
    assert A.shape == (12, 12)
    w, vt = np.zeros(12), np.zeros((12, 12))
    lib().orc_eig12(_p(A, c_dp), _p(w, c_dp), _p(vt, c_dp))
    return w, vt


def eig12_low4(A):
    """The specification's solver since round 4 (epnp.inc::epnp_eig12_low4): the four smallest eigenvalues (ascending) of the
    symmetric 12 x 12 A and their eigenvectors (rows of v4) — tridiagonalisation, bisection, inverse iteration, back-transformation."""
    A = _d(A)
    assert A.shape == (12, 12)
    w4, v4 = np.zeros(4), np.zeros((4, 12))
    lib().orc_eig12_low4(_p(A, c_dp), _p(w4, c_dp), _p(v4, c_dp))
    return w4, v4


def set_epnp_eig_mode(mode):
    """Which eigen-solver EPnP uses for M^T M: 0 / False = the specification (eig12_low4); 1 / True = cyclic Jacobi, 2 = round 3's
    Householder + implicit QL (eig12) — complete decompositions, kept as cross-checks of the tests."""
    lib().orc_set_epnp_eig_mode(ctypes.c_int(int(mode)))


def set_epnp_refit_f64(on):
    """Version-dependent decision (i) of epnp.inc: True (default) = solvePnPRansac's re-fit sees float64 normalised image points."""
    lib().orc_set_epnp_refit_f64(ctypes.c_int(1 if on else 0))


def set_cov_waves(waves):
    """Summation tree of the covariance Hessian in u2d_pnp / u2d_pnp_epnp: the kernel's for `waves` waves per object (default 4 = what
    the library launches for fewer than 2048 objects; 2 beyond; mr_pick_waves) — orc_cov_hessian_spec —, or 0 = sequential over the
    points (rounds 1-5; orc_torch_jacobian)."""
    lib().orc_set_cov_waves(ctypes.c_int(int(waves)))


def cov_hessian_spec(K, z_min, u_range, v_range, yaw, t, x3d, istd, inlier, waves=4):
    """The specified covariance Hessian of ONE object (orc_cov_hessian_spec): (4,4) float64."""
    K, ur, vr, t = _d(K).reshape(9), _d(u_range).reshape(2), _d(v_range).reshape(2), _d(t).reshape(3)
    x3d, istd = _d(x3d), _d(istd)
    m = np.ascontiguousarray(np.asarray(inlier).astype(np.uint8)) if inlier is not None else None
    H = np.zeros((4, 4))
    lib().orc_cov_hessian_spec(_p(K, c_dp), ctypes.c_double(z_min), _p(ur, c_dp), _p(vr, c_dp), ctypes.c_double(float(yaw)), _p(t, c_dp), _p(x3d, c_dp), _p(istd, c_dp),
                               _p(m, c_u8p) if m is not None else None, ctypes.c_int(x3d.shape[0]), ctypes.c_int(int(waves)), _p(H, c_dp))
    return H


def spec_sincos(x):
    """The covariance stage's specified sin / cos (orc_spec_sincos: the twin of csrc/pnp_kernel.inc spec_sincos), element-wise."""
    x = np.atleast_1d(_d(x))
    sn, cs = np.zeros_like(x), np.zeros_like(x)
    a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
    for i, v in enumerate(x):
        lib().orc_spec_sincos_export(ctypes.c_double(float(v)), ctypes.byref(a), ctypes.byref(b))
        sn[i], cs[i] = a.value, b.value
    return sn, cs


def set_epnp_moments(on):
    """True (default) = the n-point EPnP systems (re-fit on the inliers, plain solvePnP) build M^T M and the absolute orientation from
    moment sums, the kernel's form since round 5; False = entry by entry / two passes per candidate (round 4's form): the independent
    check of that reformulation."""
    lib().orc_set_epnp_moments(ctypes.c_int(1 if on else 0))


def set_epnp_cv_early_return(on):
    """Version-dependent decision (ii) of epnp.inc: True = OpenCV >= 3.3's early return for exactly five candidates (EPnP on the float32
    inputs); False (default) = the float64 normalisation of every re-fit.  Four candidates: EPnP either way (P3P is not restated)."""
    lib().orc_set_epnp_cv_early_return(ctypes.c_int(1 if on else 0))


def set_lm_iter0_gradient_test(on):
    """Version-dependent decision (iii): True (default) = Ceres tests the gradient tolerance before the first step."""
    lib().orc_set_lm_iter0_gradient_test(ctypes.c_int(1 if on else 0))


def svd_small(A):
    A = _d(A); m, n = A.shape
    w, u, v = np.zeros(n), np.zeros((m, n)), np.zeros((n, n))
    lib().orc_svd_small(ctypes.c_int(m), ctypes.c_int(n), _p(A, c_dp), _p(w, c_dp), _p(u, c_dp), _p(v, c_dp))
    return w, u, v


def cv_rng_uniform(seed, count, a, b):
    """`count` draws of cv::RNG(seed).uniform(a, b) (the generator RANSAC samples its subsets with)."""
    out = np.zeros(count, np.int32)
    lib().orc_cv_rng(ctypes.c_uint64(seed), ctypes.c_int(count), ctypes.c_int(a), ctypes.c_int(b), _p(out, c_ip))
    return out


def pnp6_uncert(pts2d, pts3d, wgt2d, K, init_pose6, clips):
    """6-DoF solve of one object (host fp64): returns dict(val, pose (6,), cov (6,6), tr, iters, why)."""
    pts2d, pts3d, wgt2d, K, init_pose6, clips = map(_d, (pts2d, pts3d, wgt2d, K, init_pose6, clips))
    val, pose, cov, tr, diag = np.zeros(1, np.int32), np.zeros(6), np.eye(6), np.zeros(1), np.zeros(5)
    lib().orc_pnp6_uncert(_p(pts2d, c_dp), _p(pts3d, c_dp), _p(wgt2d, c_dp), _p(K, c_dp), _p(init_pose6, c_dp), _p(val, c_ip), _p(pose, c_dp),
                          _p(cov, c_dp), _p(tr, c_dp), ctypes.c_int(pts2d.shape[0]), _p(clips, c_dp), _p(diag, c_dp))
    return dict(val=int(val[0]), pose=pose, cov=cov, tr=float(tr[0]), iters=int(diag[0]), why=int(diag[1]), final_cost=diag[4])


def eval6(pts2d, pts3d, wgt2d, K, pose6, clips):
    """cost, gradient (6,), J^T J (6,6), residuals (pn,2), Jacobian (pn,2,6) of the 6-DoF problem at pose6 (Ceres / Jet semantics)."""
    pts2d, pts3d, wgt2d, K, pose6, clips = map(_d, (pts2d, pts3d, wgt2d, K, pose6, clips))
    pn = pts2d.shape[0]
    cost, g, H, res, jac = np.zeros(1), np.zeros(6), np.zeros((6, 6)), np.zeros((pn, 2)), np.zeros((pn, 2, 6))
    f = lib().orc_eval6
    f.restype = ctypes.c_int
    ok = f(_p(pts2d, c_dp), _p(pts3d, c_dp), _p(wgt2d, c_dp), _p(K, c_dp), _p(pose6, c_dp), ctypes.c_int(pn), _p(clips, c_dp), _p(cost, c_dp), _p(g, c_dp),
           _p(H, c_dp), _p(res, c_dp), _p(jac, c_dp))
    return bool(ok), float(cost[0]), g, H, res, jac


def pnp6_refine(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, mask, pose4, valid4, z_min=0.5, num_threads=1):
    """Batch 6-DoF refinement from the 4-DoF result on its inlier mask: returns (valid (B,), pose6 (B,6) f32, cov6 (B,6,6) f32, diag (B,2))."""
    B, P = coords_2d.shape[:2]
    x2d, istd, x3d = _f(coords_2d), _f(coords_2d_istd), _f(coords_3d)
    K = _f(cam_mats).reshape(-1, 9); ur, vr = _f(u_range).reshape(-1, 2), _f(v_range).reshape(-1, 2)
    m = np.ascontiguousarray(mask, np.uint8); p4 = _f(pose4).reshape(B, 4); v4 = np.ascontiguousarray(valid4, np.uint8)
    valid, pose6, cov6, diag = np.zeros(B, np.uint8), np.zeros((B, 6), np.float32), np.zeros((B, 36), np.float32), np.zeros((B, 2), np.float32)
    lib().orc_pnp6_refine_batch(_p(x2d, c_fp), _p(istd, c_fp), _p(x3d, c_fp), _p(K, c_fp), ctypes.c_int(K.shape[0]), _p(ur, c_fp), _p(vr, c_fp),
                                ctypes.c_int(ur.shape[0]), _p(m, c_u8p), _p(p4, c_fp), _p(v4, c_u8p), ctypes.c_int(B), ctypes.c_int(P),
                                ctypes.c_double(z_min), ctypes.c_int(num_threads), _p(valid, c_u8p), _p(pose6, c_fp), _p(cov6, c_fp), _p(diag, c_fp))
    return valid.astype(bool), pose6, cov6.reshape(B, 6, 6), diag


def max_threads():
    return int(lib().orc_max_threads())


# ------------------------------------------------------------------ numpy restatements (fp32) ---
NOC_MEANS = np.array((-0.1, -0.5, 0.0), np.float32)      # noc_coder.py:9
NOC_STDS = np.array((0.35, 0.23, 0.34), np.float32)      # noc_coder.py:10
DIM_MEANS = np.array([(3.89, 1.53, 1.62), (0.82, 1.78, 0.63), (1.77, 1.72, 0.57)], np.float32)  # multiclass_norm_dim_coder.py:8-11
DIM_STDS = np.array([(0.44, 0.14, 0.11), (0.25, 0.13, 0.12), (0.15, 0.10, 0.14)], np.float32)   # :12-15


def slice_pred(all_pred, labels, flip, num_classes=3, class_agnostic=False):
    """R9: flip-branch select (fcn_noc_decoder.py:225-235) + class channel gather (:242-267).
    all_pred (B, 2*C*5, h, w) -> noc (B,3,h,w), logstd (B,2,h,w).  Also returns the int channel
    table (B,5): channel of noc comp k = f*5C + 3c + k ; log-std comp k = f*5C + 3C + 2c + k."""
    B, ch, h, w = all_pred.shape
    C = 1 if class_agnostic else num_classes
    assert ch == 2 * C * 5
    flip = np.broadcast_to(np.asarray(flip, bool), (B,)).astype(np.int64)
    labels = np.zeros(B, np.int64) if class_agnostic else np.asarray(labels, np.int64)
    chan = np.empty((B, 5), np.int64)
    for k in range(3):
        chan[:, k] = flip * 5 * C + 3 * labels + k
    for k in range(2):
        chan[:, 3 + k] = flip * 5 * C + 3 * C + 2 * labels + k
    ar = np.arange(B)
    noc = np.stack([all_pred[ar, chan[:, k]] for k in range(3)], axis=1)
    logstd = np.stack([all_pred[ar, chan[:, 3 + k]] for k in range(2)], axis=1)
    return noc, logstd, chan


def dim_decode(dim, dim_var, labels):
    """R10a (multiclass_norm_dim_coder.py:28-36)."""
    mu, sd = DIM_MEANS[labels], DIM_STDS[labels]
    dims = dim * sd + mu
    return dims, (dim_var * np.square(sd) if dim_var is not None else None)


def noc_decode(noc, dims, dims_var):
    """R10b (noc_coder.py:50-73) for the test-time case noc_var=None."""
    part = noc * NOC_STDS[:, None, None] + NOC_MEANS[:, None, None]
    c3d = part * dims[..., None, None]
    var = dims_var[..., None, None] * np.square(part) if dims_var is not None else None
    return c3d, var


def spec_expf(x):
    """float32 exp as SPECIFIED for the HIP decode (monorun_pnp.hip::mr_expf): the same sequence of IEEE float32 multiplications
    and additions (numpy float32 arithmetic is exactly that), so the result is bit-identical to the kernel's."""
    x = np.asarray(x, np.float32)
    f = np.float32
    with np.errstate(over='ignore', under='ignore', invalid='ignore'):
        kf = np.rint(x * f(1.44269504088896341))
        r = x - kf * f(0.693359375)
        r = r - kf * f(-2.12194440e-4)
        z = r * r
        p = f(1.9875691500E-4) * r + f(1.3981999507E-3)
        p = p * r + f(8.3334519073E-3)
        p = p * r + f(4.1665795894E-2)
        p = p * r + f(1.6666665459E-1)
        p = p * r + f(5.0000001201E-1)
        y = p * z + r
        y = y + f(1.0)
        k = np.where(np.isfinite(kf), kf, 0).astype(np.int32)
        out = np.ldexp(y, k).astype(np.float32)
    out = np.where(x > f(88.72283935546875), f(np.inf), out)
    out = np.where(x < f(-103.0), f(0.0), out)
    return out.astype(np.float32)


def spec_logf(x):
    """float32 log as SPECIFIED for the HIP decode (monorun_pnp.hip::mr_logf), operation for operation."""
    x = np.asarray(x, np.float32)
    f = np.float32
    with np.errstate(all='ignore'):
        m, e = np.frexp(x)
        m = m.astype(np.float32)
        lo = m < f(0.707106781186547524)
        e = np.where(lo, e - 1, e)
        m = np.where(lo, m + m - f(1.0), m - f(1.0)).astype(np.float32)
        z = m * m
        p = f(7.0376836292E-2) * m - f(1.1514610310E-1)
        p = p * m + f(1.1676998740E-1)
        p = p * m - f(1.2420140846E-1)
        p = p * m + f(1.4249322787E-1)
        p = p * m - f(1.6668057665E-1)
        p = p * m + f(2.0000714765E-1)
        p = p * m - f(2.4999993993E-1)
        p = p * m + f(3.3333331174E-1)
        fe = e.astype(np.float32)
        y = m * (z * p)
        y = y + f(-2.12194440e-4) * fe
        y = y - f(0.5) * z
        zz = m + y
        out = (zz + f(0.693359375) * fe).astype(np.float32)
    out = np.where(x == f(np.inf), f(np.inf), out)
    out = np.where(x == 0, f(-np.inf), out)
    out = np.where(~(x >= 0), f(np.nan), out)
    return out.astype(np.float32)


def decode_logstd(proj_logstd, c3d_var, ref_length=1.6, ref_focal_y=722, target_std=0.15, epistemic_std_gain=1.0, exp=np.exp, log=np.log):
    """R11 (distance_invar_proj_error_coder.py:39-60) with distance=None."""
    sd_py = ref_length * ref_focal_y * target_std          # Python float: torch rounds `tensor * python_scalar` operands to float32
    sd = np.float32(sd_py)                                 # new_tensor([scaling_denomitor]) (:41-42)
    if c3d_var is None:
        return proj_logstd + np.log(sd / sd)
    v2 = np.empty(proj_logstd.shape, np.float32)
    v2[:, 0] = np.float32(0.5) * (c3d_var[:, 0] + c3d_var[:, 2])
    v2[:, 1] = c3d_var[:, 1]
    v2 = (v2 * np.float32((ref_focal_y * epistemic_std_gain) ** 2)
          + exp(np.float32(2) * proj_logstd) * np.float32(sd_py ** 2)) / np.square(sd)
    return (np.float32(0.5) * log(v2)).astype(np.float32)


def roi_grid(rois_xyxy, h=28, w=28):
    """R12, interior analytic form of roi_align(coord_2d, rois, (h,w), 1.0, 0, 'avg', aligned=True):
    u(px) = x1 - 0.5 + (px + 0.5) * (x2 - x1) / w  (mmcv absent -> border unpinned)."""
    r = np.asarray(rois_xyxy, np.float32)
    x1, y1, x2, y2 = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    px = (np.arange(w, dtype=np.float32) + np.float32(0.5))
    py = (np.arange(h, dtype=np.float32) + np.float32(0.5))
    u = (x1 - np.float32(0.5))[:, None] + px[None, :] * ((x2 - x1) / np.float32(w))[:, None]
    v = (y1 - np.float32(0.5))[:, None] + py[None, :] * ((y2 - y1) / np.float32(h))[:, None]
    out = np.empty((r.shape[0], 2, h, w), np.float32)
    out[:, 0] = u[:, None, :]
    out[:, 1] = v[:, :, None]
    return out


def pose_head_prep(coords_2d, coords_2d_logstd, coords_3d, img_shapes, allowed_border=200,
                   epnp_ransac_thres_ratio=0.2, std_scale=10, exp=np.exp):
    """R8 (uncert_prop_pnp_optimizer.py:73-88): NCHW maps -> the PnP boundary tensors, keeping the
    reference's *strided views* (permute(0,2,3,1).view -> strides (C*hw, 1, hw))."""
    bn, _, h, w = coords_2d.shape
    istd = (exp(-coords_2d_logstd) / np.float32(std_scale)).astype(np.float32)
    img_shapes = np.asarray(img_shapes, np.float32).reshape(-1, 2)
    u_range = np.full((img_shapes.shape[0], 2), -allowed_border, np.float32)
    v_range = np.full((img_shapes.shape[0], 2), -allowed_border, np.float32)
    u_range[:, 1] = img_shapes[:, 1] + allowed_border
    v_range[:, 1] = img_shapes[:, 0] + allowed_border
    def pv(a):
        return a.reshape(bn, a.shape[1], h * w).transpose(0, 2, 1)
    roi_h = coords_2d[:, 1, -1, 0] - coords_2d[:, 1, 0, 0]
    thr = (np.float32(epnp_ransac_thres_ratio) * roi_h).astype(np.float32) if epnp_ransac_thres_ratio is not None else None
    return pv(coords_2d), pv(istd), pv(coords_3d), u_range, v_range, thr


def cov_calib(pose_cov, cov_calib_logscale):
    """uncert_prop_pnp_optimizer.py:96-97."""
    s = np.exp(np.asarray(cov_calib_logscale, np.float32))
    return (s * s[:, None]) * pose_cov


def cov_correction(cov, t_vec, ref_length=1.6, ref_focal_y=722, target_std=0.15):
    """R13 (distance_invar_proj_error_coder.py:62-63 with 'range' distance, uncert_projection_head.py:104-109)."""
    sd = np.float32(ref_length * ref_focal_y * target_std)
    dist = np.linalg.norm(t_vec.astype(np.float32), axis=1)
    return cov * np.square(sd / dist).reshape(-1, 1, 1)


# ------------------------------------------------------------------ N1: consumers of the pose -----
# Rotated-BEV NMS.  The reference calls mmdet3d.ops.iou3d.nms_gpu (third-party, not in the tree, version
# pinned only by INSTALL.md's mmdet3d 0.8.0 line) at monorun_roi_head.py:638-639 on boxes
# [x1, y1, x2, y2, ry] built by xywhr2xyxyr (:657-677).  Restated from the published algorithm of
# mmdet3d/ops/iou3d (sort by score, pairwise rotated-rectangle IoU by polygon clipping, greedy suppression
# of IoU > thr, IoU = overlap / max(area_a + area_b - overlap, 1e-8)); fp64, Sutherland-Hodgman clipping.
def _bev_corners(b):
    x1, y1, x2, y2, ang = [float(v) for v in b]
    cx, cy, hw, hh = 0.5 * (x1 + x2), 0.5 * (y1 + y2), 0.5 * (x2 - x1), 0.5 * (y2 - y1)
    c, s = np.cos(ang), np.sin(ang)
    pts = []
    for dx, dy in ((-hw, -hh), (hw, -hh), (hw, hh), (-hw, hh)):
        # mmdet3d rotate_around_center: x' = dx*cos + dy*sin, y' = -dx*sin + dy*cos
        pts.append((cx + dx * c + dy * s, cy - dx * s + dy * c))
    return pts


def _poly_area(p):
    a = 0.0
    for i in range(len(p)):
        x0, y0 = p[i]
        x1, y1 = p[(i + 1) % len(p)]
        a += x0 * y1 - x1 * y0
    return 0.5 * a


def _clip(subject, clipper):
    """Sutherland-Hodgman: clip polygon `subject` by convex polygon `clipper` (both CCW)."""
    out = subject
    for i in range(len(clipper)):
        ax, ay = clipper[i]
        bx, by = clipper[(i + 1) % len(clipper)]
        inp, out = out, []
        if not inp:
            break
        def side(p):
            return (bx - ax) * (p[1] - ay) - (by - ay) * (p[0] - ax)
        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp >= 0) != (sq >= 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def rotated_iou_bev(a, b):
    pa, pb = _bev_corners(a), _bev_corners(b)
    if _poly_area(pa) < 0:
        pa = pa[::-1]
    if _poly_area(pb) < 0:
        pb = pb[::-1]
    inter = _clip(pa, pb)
    ov = abs(_poly_area(inter)) if len(inter) >= 3 else 0.0
    sa = abs((a[2] - a[0]) * (a[3] - a[1]))
    sb = abs((b[2] - b[0]) * (b[3] - b[1]))
    return ov / max(sa + sb - ov, 1e-8)


def nms_bev(boxes_xyxyr, scores, thr):
    """Greedy rotated NMS; returns kept indices into the input, in descending-score order
    (ties: lower index first)."""
    boxes = np.asarray(boxes_xyxyr, np.float64)
    scores = np.asarray(scores, np.float64)
    order = sorted(range(len(scores)), key=lambda i: (-scores[i], i))
    keep, dead = [], set()
    for ii, i in enumerate(order):
        if i in dead:
            continue
        keep.append(i)
        for j in order[ii + 1:]:
            if j not in dead and rotated_iou_bev(boxes[i], boxes[j]) > thr:
                dead.add(j)
    return np.array(keep, np.int64)


def xywhr2xyxyr(b):
    """monorun_roi_head.py:657-677."""
    b = np.asarray(b)
    out = np.zeros_like(b)
    out[:, 0] = b[:, 0] - b[:, 2] / 2
    out[:, 1] = b[:, 1] - b[:, 3] / 2
    out[:, 2] = b[:, 0] + b[:, 2] / 2
    out[:, 3] = b[:, 1] + b[:, 3] / 2
    out[:, 4] = b[:, 4]
    return out


def score_head_inputs(yaw, t_vec, pose_cov, dims):
    """mlp_score_head.py:101-103: [yaw, t, tril(cov) in torch.tril_indices(4,4) order, dims] -> (n, 17)."""
    r, c = np.tril_indices(4)
    return np.concatenate([yaw, t_vec, pose_cov[:, r, c], dims], axis=1)


# ------------------------------------------------------------------ N4: the 7-parameter variants ---
def pnp_noc(pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, init_dimpose, clips, delta, full_cov=False):
    """pnp_noc_uncert / pnp_noc_cov_uncert (ext.h:15-43): returns dict(val, dimpose, iters, why, ...)."""
    pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, init_dimpose, clips = map(_d, (pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, init_dimpose, clips))
    val = np.zeros(1, np.int32)
    out = np.zeros(7)
    diag = np.zeros(6)
    lib().orc_noc_solve_diag(ctypes.c_int(int(full_cov)), _p(pts2d, c_dp), _p(pts3d, c_dp), _p(wgt2d, c_dp), _p(logdim, c_dp),
                             _p(logdim_wgt, c_dp), _p(K, c_dp), _p(init_dimpose, c_dp), _p(val, c_ip), _p(out, c_dp),
                             ctypes.c_int(pts2d.shape[0]), _p(clips, c_dp), ctypes.c_double(delta), _p(diag, c_dp))
    return dict(val=int(val[0]), dimpose=out, iters=int(diag[0]), why=int(diag[1]), termination=int(diag[2]),
                initial_cost=diag[3], final_cost=diag[4])


def noc_cost_grad(pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, x, clips, delta, full_cov=False):
    """Robustified cost 1/2 sum rho(|r_block|^2), its gradient J^T r and J^T J (corrected) at x."""
    pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, x, clips = map(_d, (pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, x, clips))
    cost = np.zeros(1); g = np.zeros(7); H = np.zeros((7, 7))
    lib().orc_noc_cost_grad.restype = ctypes.c_int
    ok = lib().orc_noc_cost_grad(ctypes.c_int(int(full_cov)), _p(pts2d, c_dp), _p(pts3d, c_dp), _p(wgt2d, c_dp), _p(logdim, c_dp),
                                 _p(logdim_wgt, c_dp), _p(K, c_dp), _p(x, c_dp), ctypes.c_int(pts2d.shape[0]), _p(clips, c_dp),
                                 ctypes.c_double(delta), _p(cost, c_dp), _p(g, c_dp), _p(H, c_dp))
    return bool(ok), float(cost[0]), g, H


# ------------------------------------------------------------------ N3: RoIAlign (average pooling) ---
def roi_align_avg(inp, rois, out_hw, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    """mmcv.ops.roi_align(input, rois, out_hw, spatial_scale, sampling_ratio, 'avg', aligned) forward, float32, operation
    for operation as the published kernel (mmcv 1.2.1 roi_align_cuda_kernel.cuh; mmcv is third-party and absent from the
    reference tree: parity with it is UNPINNED, the restatement is checked against closed forms in tests/).
    inp (N,C,H,W), rois (K,5) [batch_idx, x1, y1, x2, y2] -> (K,C,oh,ow)."""
    f = np.float32
    inp = np.asarray(inp, f)
    rois = np.asarray(rois, f)
    N, C, H, W = inp.shape
    oh, ow = out_hw
    out = np.zeros((rois.shape[0], C, oh, ow), f)
    ph = np.arange(oh, dtype=f)[:, None]
    pw = np.arange(ow, dtype=f)[None, :]
    for n in range(rois.shape[0]):
        bi = int(rois[n, 0])
        off = f(0.5) if aligned else f(0.0)
        sw, sh = f(rois[n, 1] * f(spatial_scale)) - off, f(rois[n, 2] * f(spatial_scale)) - off
        rw, rh = (f(rois[n, 3] * f(spatial_scale)) - off) - sw, (f(rois[n, 4] * f(spatial_scale)) - off) - sh
        if not aligned:
            rw, rh = max(rw, f(1.0)), max(rh, f(1.0))
        bh, bw = f(rh / f(oh)), f(rw / f(ow))
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(f(rh / f(oh))))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(f(rw / f(ow))))
        count = f(max(gh * gw, 1))
        acc = np.zeros((C, oh, ow), f)
        for iy in range(gh):
            y = (sh + ph * bh) + f(f(f(iy) + f(0.5)) * bh) / f(gh)                  # (oh,1)
            for ix in range(gw):
                x = (sw + pw * bw) + f(f(f(ix) + f(0.5)) * bw) / f(gw)              # (1,ow)
                yy = np.broadcast_to(y, (oh, ow)).astype(f).copy()
                xx = np.broadcast_to(x, (oh, ow)).astype(f).copy()
                dead = (yy < -1.0) | (yy > H) | (xx < -1.0) | (xx > W)
                yy = np.where(yy <= 0, f(0), yy)
                xx = np.where(xx <= 0, f(0), xx)
                yl, xl = yy.astype(np.int32), xx.astype(np.int32)
                top, right = yl >= H - 1, xl >= W - 1
                yl = np.where(top, H - 1, yl); xl = np.where(right, W - 1, xl)
                yh = np.where(top, H - 1, yl + 1); xh = np.where(right, W - 1, xl + 1)
                yy = np.where(top, yl.astype(f), yy); xx = np.where(right, xl.astype(f), xx)
                ly, lx = (yy - yl.astype(f)).astype(f), (xx - xl.astype(f)).astype(f)
                hy, hx = (f(1.0) - ly).astype(f), (f(1.0) - lx).astype(f)
                m = inp[bi]
                v1, v2, v3, v4 = m[:, yl, xl], m[:, yl, xh], m[:, yh, xl], m[:, yh, xh]
                val = (((hy * hx) * v1 + (hy * lx) * v2).astype(f) + (ly * hx) * v3).astype(f) + (ly * lx) * v4
                acc = (acc + np.where(dead, f(0), val.astype(f))).astype(f)
        out[n] = acc / count
    return out

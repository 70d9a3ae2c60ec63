"""Torch-facing uncertainty-aware PnP op — drop-in for
/root/reference/monorun/ops/least_squares/pnp_uncert.py (``pnp_uncert`` :7-87, ``PnPUncert`` :90-142).

Same names, argument meaning, return tuple, dtypes and devices.  What differs is the execution:
the reference copies six tensors to the host, loops over objects in Python (cv2 EPnP/RANSAC + cffi
Ceres LM), copies back and builds J^T J with ~40 small torch kernels; here ONE fused HIP kernel
(``mr_pnp_uncert_batched`` in include/monorun_pnp.h) does mask -> initialiser -> LM -> covariance on the
device, on the caller's stream, with no host synchronisation.
"""
import ctypes

import torch

from ... import _lib
from .builder import PNP

_DTYPES = {torch.float32: _lib.MR_F32, torch.float16: _lib.MR_F16, torch.float64: _lib.MR_F64}

# What `build_pnp(dict(type='PnPUncert', ...))` from the reference's own config dict, `pnp_uncert`, `u2d_pnp_cpu` and the pose head run when
# nobody says otherwise: the REFERENCE's flow — cv2.solvePnPRansac(EPNP, 30 iterations) restated on the GPU, then the LM + covariance
# (pnp_uncert_cpu.py:33-68).  'k0' (this repository's one-launch consensus initialiser: ~5 x the throughput, the same LM, inlier sets that
# differ from the reference flow's on ~13 % of the objects) is an explicit fast mode since round 5.
DEFAULT_INITIALISER = 'epnp'


def _strides(t):
    return (ctypes.c_int64 * 3)(*t.stride())


def pnp_uncert_device(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=0.5,
                      epnp_istd_thres=1.0, epnp_ransac_thres=None, inlier_opt_only=False,
                      init_pose=None, flags=0, with_diag=False):
    """Launch the fused kernel on CUDA(HIP) tensors; returns raw device outputs
    (valid u8 (B,), pose f32 (B,4), cov f32 (B,4,4), tr f32 (B,), mask u8 (B,P), diag f32 (B,4)|None)."""
    lib = _lib.load()
    dev = coords_2d.device
    if dev.type != 'cuda':
        raise RuntimeError('monorun_amd PnP runs on an MI355X only: inputs must be on a HIP device '
                           '(there is no CPU fallback)')
    B, P = int(coords_2d.shape[0]), int(coords_2d.shape[1])
    dt = coords_2d.dtype if coords_2d.dtype in _DTYPES else torch.float32

    def prep(t):
        t = t.detach()
        return t if (t.dtype == dt and t.device == dev) else t.to(device=dev, dtype=dt)
    x2d, istd, x3d = prep(coords_2d), prep(coords_2d_istd), prep(coords_3d)
    assert x2d.shape == (B, P, 2) and istd.shape == (B, P, 2) and x3d.shape == (B, P, 3)
    f32 = dict(device=dev, dtype=torch.float32)
    cam = cam_mats.detach().to(**f32).reshape(-1, 3, 3).contiguous()
    ur = u_range.detach().to(**f32).reshape(-1, 2).contiguous()
    vr = v_range.detach().to(**f32).reshape(-1, 2).contiguous()
    assert ur.shape[0] == vr.shape[0]
    thr = epnp_ransac_thres.detach().to(**f32).reshape(-1).contiguous() if epnp_ransac_thres is not None else None
    ini = init_pose.detach().to(device=dev, dtype=torch.float64).reshape(-1, 4).contiguous() if init_pose is not None else None
    valid = torch.empty(B, device=dev, dtype=torch.uint8)
    pose = torch.empty(B, 4, **f32)
    cov = torch.empty(B, 4, 4, **f32)
    tr = torch.empty(B, **f32)
    mask = torch.empty(B, P, device=dev, dtype=torch.uint8)
    diag = torch.empty(B, 4, **f32) if with_diag else None
    if B > 0:
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.mr_pnp_uncert_batched(
                x2d.data_ptr(), _strides(x2d), istd.data_ptr(), _strides(istd), x3d.data_ptr(), _strides(x3d), _DTYPES[dt],
                cam.data_ptr(), cam.shape[0], ur.data_ptr(), vr.data_ptr(), ur.shape[0],
                thr.data_ptr() if thr is not None else None, ini.data_ptr() if ini is not None else None, B, P,
                float(z_min), float(epnp_istd_thres), int(bool(inlier_opt_only)), int(flags),
                valid.data_ptr(), pose.data_ptr(), cov.data_ptr(), tr.data_ptr(), mask.data_ptr(),
                diag.data_ptr() if diag is not None else None, stream))
    return valid, pose, cov, tr, mask, diag


def epnp_ransac_device(coords_2d, coords_2d_istd, coords_3d, cam_mats, epnp_istd_thres=1.0, epnp_ransac_thres=None, flags=0,
                       max_iters=30, with_diag=False, debug_hypotheses=False, first_round=None):
    """The reference's own initialiser on the GPU (``mr_epnp_ransac_batched``): cv2.solvePnPRansac(..., iterationsCount=30,
    flags=SOLVEPNP_EPNP) on the istd candidates of every object (plain EPnP without thresholds), pnp_uncert_cpu.py:33-68.
    Returns (init_pose f64 (B,4) [yaw0, t], init_mask u8 (B,P), init_valid u8 (B,), diag f32 (B,4)|None, hypotheses f64 (B,30,12)|None)."""
    lib = _lib.load()
    dev = coords_2d.device
    if dev.type != 'cuda':
        raise RuntimeError('monorun_amd EPnP/RANSAC runs on an MI355X only (no CPU fallback)')
    if first_round is not None:                       # hypotheses solved for every object before the replayed loop is consulted (1..30; result-neutral)
        flags = (int(flags) & ~(0x1F << _lib.MR_EPNP_FIRST_ROUND_SHIFT)) | (max(1, min(30, int(first_round))) << _lib.MR_EPNP_FIRST_ROUND_SHIFT)
    B, P = int(coords_2d.shape[0]), int(coords_2d.shape[1])
    dt = coords_2d.dtype if coords_2d.dtype in _DTYPES else torch.float32
    prep = lambda t: t.detach() if (t.dtype == dt and t.device == dev) else t.detach().to(device=dev, dtype=dt)
    x2d, istd, x3d = prep(coords_2d), prep(coords_2d_istd), prep(coords_3d)
    f32 = dict(device=dev, dtype=torch.float32)
    cam = cam_mats.detach().to(**f32).reshape(-1, 3, 3).contiguous()
    thr = epnp_ransac_thres.detach().to(**f32).reshape(-1).contiguous() if epnp_ransac_thres is not None else None
    init_pose = torch.empty(B, 4, device=dev, dtype=torch.float64)
    init_mask = torch.empty(B, P, device=dev, dtype=torch.uint8)
    init_valid = torch.empty(B, device=dev, dtype=torch.uint8)
    diag = torch.empty(B, 4, **f32) if with_diag else None
    hyp = torch.zeros(B, 30, 12, device=dev, dtype=torch.float64) if debug_hypotheses else None
    if B > 0:
        with torch.cuda.device(dev):
            # the launches of the call hand their intermediate results over in a workspace: from torch's caching allocator, on the
            # stream the launches go to (the block returns to the allocator when `work` dies; stream order keeps that safe)
            work = torch.empty(int(lib.mr_epnp_workspace_bytes(B, P)), device=dev, dtype=torch.uint8)
            _lib.check(lib.mr_epnp_ransac_batched(
                x2d.data_ptr(), _strides(x2d), istd.data_ptr(), _strides(istd), x3d.data_ptr(), _strides(x3d), _DTYPES[dt],
                cam.data_ptr(), cam.shape[0], thr.data_ptr() if thr is not None else None, B, P, float(epnp_istd_thres), int(flags), int(max_iters),
                init_pose.data_ptr(), init_mask.data_ptr(), init_valid.data_ptr(), diag.data_ptr() if diag is not None else None,
                hyp.data_ptr() if hyp is not None else None, work.data_ptr(), work.numel(), torch.cuda.current_stream(dev).cuda_stream))
    return init_pose, init_mask, init_valid, diag, hyp


def pnp_uncert_from_init_device(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, init_pose, init_mask, init_valid,
                                z_min=0.5, inlier_opt_only=False, flags=0, with_diag=False):
    """LM + covariance from an external initialiser's (init_pose f64 (B,4), init_mask u8 (B,P), init_valid u8 (B,))
    (``mr_pnp_uncert_from_init_batched``).  Returns what ``pnp_uncert_device`` returns."""
    lib = _lib.load()
    dev = coords_2d.device
    B, P = int(coords_2d.shape[0]), int(coords_2d.shape[1])
    dt = coords_2d.dtype if coords_2d.dtype in _DTYPES else torch.float32
    prep = lambda t: t.detach() if (t.dtype == dt and t.device == dev) else t.detach().to(device=dev, dtype=dt)
    x2d, istd, x3d = prep(coords_2d), prep(coords_2d_istd), prep(coords_3d)
    f32 = dict(device=dev, dtype=torch.float32)
    cam = cam_mats.detach().to(**f32).reshape(-1, 3, 3).contiguous()
    ur = u_range.detach().to(**f32).reshape(-1, 2).contiguous()
    vr = v_range.detach().to(**f32).reshape(-1, 2).contiguous()
    ini = init_pose.detach().to(device=dev, dtype=torch.float64).reshape(-1, 4).contiguous()
    im = init_mask.detach().to(device=dev, dtype=torch.uint8).contiguous()
    iv = init_valid.detach().to(device=dev, dtype=torch.uint8).contiguous()
    valid = torch.empty(B, device=dev, dtype=torch.uint8)
    pose = torch.empty(B, 4, **f32)
    cov = torch.empty(B, 4, 4, **f32)
    tr = torch.empty(B, **f32)
    mask = torch.empty(B, P, device=dev, dtype=torch.uint8)
    diag = torch.empty(B, 4, **f32) if with_diag else None
    if B > 0:
        with torch.cuda.device(dev):
            _lib.check(lib.mr_pnp_uncert_from_init_batched(
                x2d.data_ptr(), _strides(x2d), istd.data_ptr(), _strides(istd), x3d.data_ptr(), _strides(x3d), _DTYPES[dt],
                cam.data_ptr(), cam.shape[0], ur.data_ptr(), vr.data_ptr(), ur.shape[0],
                ini.data_ptr(), im.data_ptr(), iv.data_ptr(), B, P, float(z_min), int(bool(inlier_opt_only)), int(flags),
                valid.data_ptr(), pose.data_ptr(), cov.data_ptr(), tr.data_ptr(), mask.data_ptr(),
                diag.data_ptr() if diag is not None else None, torch.cuda.current_stream(dev).cuda_stream))
    return valid, pose, cov, tr, mask, diag


def pnp_uncert_epnp_device(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=0.5, epnp_istd_thres=1.0,
                           epnp_ransac_thres=None, inlier_opt_only=False, flags=0, max_iters=30, with_diag=False, first_round=None):
    """The reference's flow in one call: ``epnp_ransac_device`` with MR_EPNP_DEFER_REFIT, then ``mr_pnp_uncert_from_epnp_grouped`` — the LM +
    covariance launch that also runs the initialiser's last step (the re-fit's pose candidates) on the tile it loads anyway.  Same results,
    bit for bit, as ``epnp_ransac_device`` followed by ``pnp_uncert_from_init_device``; one launch and one pass over the correspondences
    less.  Returns (valid, pose, cov, tr, mask, diag|None, init_pose f64 (B,4), init_valid u8 (B,))."""
    lib = _lib.load()
    dev = coords_2d.device
    if dev.type != 'cuda':
        raise RuntimeError('monorun_amd EPnP/RANSAC runs on an MI355X only (no CPU fallback)')
    iflags = int(flags) & 0x1047                   # the bits the initialiser reads: istd mean order, MR_NO_ISTD_MASK, MR_EPNP_REFIT_F32, MR_EPNP_CV_EARLY_RETURN
    if first_round is not None:
        iflags |= max(1, min(30, int(first_round))) << _lib.MR_EPNP_FIRST_ROUND_SHIFT
    B, P = int(coords_2d.shape[0]), int(coords_2d.shape[1])
    dt = coords_2d.dtype if coords_2d.dtype in _DTYPES else torch.float32
    prep = lambda t: t.detach() if (t.dtype == dt and t.device == dev) else t.detach().to(device=dev, dtype=dt)
    x2d, istd, x3d = prep(coords_2d), prep(coords_2d_istd), prep(coords_3d)
    f32 = dict(device=dev, dtype=torch.float32)
    cam = cam_mats.detach().to(**f32).reshape(-1, 3, 3).contiguous()
    ur = u_range.detach().to(**f32).reshape(-1, 2).contiguous()
    vr = v_range.detach().to(**f32).reshape(-1, 2).contiguous()
    thr = epnp_ransac_thres.detach().to(**f32).reshape(-1).contiguous() if epnp_ransac_thres is not None else None
    init_pose = torch.empty(B, 4, device=dev, dtype=torch.float64)
    init_mask = torch.empty(B, P, device=dev, dtype=torch.uint8)
    init_valid = torch.empty(B, device=dev, dtype=torch.uint8)
    valid = torch.empty(B, device=dev, dtype=torch.uint8)
    pose = torch.empty(B, 4, **f32)
    cov = torch.empty(B, 4, 4, **f32)
    tr = torch.empty(B, **f32)
    mask = torch.empty(B, P, device=dev, dtype=torch.uint8)
    diag = torch.empty(B, 4, **f32) if with_diag else None
    if B > 0:
        one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr() if t is not None else None)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            work = torch.empty(int(lib.mr_epnp_workspace_bytes(B, P)), device=dev, dtype=torch.uint8)     # (stream order keeps its reuse safe)
            head = [x2d.data_ptr(), _strides(x2d), istd.data_ptr(), _strides(istd), x3d.data_ptr(), _strides(x3d), _DTYPES[dt], cam.data_ptr(), cam.shape[0]]
            _lib.check(lib.mr_epnp_ransac_batched(*head, thr.data_ptr() if thr is not None else None, B, P, float(epnp_istd_thres),
                                                  iflags | _lib.MR_EPNP_DEFER_REFIT, int(max_iters), init_pose.data_ptr(), init_mask.data_ptr(),
                                                  init_valid.data_ptr(), None, None, work.data_ptr(), work.numel(), st))
            _lib.check(lib.mr_pnp_uncert_from_epnp_grouped(
                1, one(x2d), _strides(x2d), one(istd), _strides(istd), one(x3d), _strides(x3d), _DTYPES[dt], one(cam), cam.shape[0], one(ur), one(vr), ur.shape[0],
                one(init_pose), one(init_mask), one(init_valid), None, B, P, float(z_min), int(bool(inlier_opt_only)), int(flags),
                one(valid), one(pose), one(cov), one(tr), one(mask), one(diag), None, 0.0, None, work.data_ptr(), work.numel(), st))
    return valid, pose, cov, tr, mask, diag, init_pose, init_valid


def cov_symeig_rule_device(valid_u8, cov, with_eigs=False):
    """The reference's eigenvalue rule (pnp_uncert.py:77-85) applied IN PLACE to (valid u8 (B,), cov f32 (B,4,4)):
    objects with lambda_min(h) <= max(1e-6 lambda_max(h), 0) become invalid and get cov = I (``mr_cov_symeig_rule``)."""
    lib = _lib.load()
    B = int(valid_u8.shape[0])
    assert valid_u8.dtype == torch.uint8 and cov.dtype == torch.float32 and cov.is_contiguous() and valid_u8.is_contiguous()
    lam = torch.empty(B, 2, device=cov.device, dtype=torch.float32) if with_eigs else None
    if B > 0:
        with torch.cuda.device(cov.device):
            _lib.check(lib.mr_cov_symeig_rule(valid_u8.data_ptr(), cov.data_ptr(), B, lam.data_ptr() if lam is not None else None,
                                              torch.cuda.current_stream(cov.device).cuda_stream))
    return lam


class PnPLaunch:
    """A prepared launch of the fused kernel over device-resident inputs with preallocated outputs:
    every ctypes argument is built once, ``run()`` only enqueues the kernel on the current stream.
    Used where the same shapes recur (bench.py, sharded serving)."""

    def __init__(self, coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=0.5,
                 epnp_istd_thres=0.6, epnp_ransac_thres=None, inlier_opt_only=True, init_pose=None, flags=0,
                 out=None, with_diag=False, mask=None):
        self.lib = _lib.load()
        dev = coords_2d.device
        if dev.type != 'cuda':
            raise RuntimeError('PnPLaunch needs HIP device tensors (no CPU fallback)')
        self.dev = dev
        B, P = int(coords_2d.shape[0]), int(coords_2d.shape[1])
        assert coords_2d.dtype in _DTYPES and coords_2d_istd.dtype == coords_2d.dtype == coords_3d.dtype
        f32 = dict(device=dev, dtype=torch.float32)
        self.keep = [coords_2d, coords_2d_istd, coords_3d,
                     cam_mats.to(**f32).reshape(-1, 3, 3).contiguous(), u_range.to(**f32).reshape(-1, 2).contiguous(),
                     v_range.to(**f32).reshape(-1, 2).contiguous(),
                     epnp_ransac_thres.to(**f32).reshape(-1).contiguous() if epnp_ransac_thres is not None else None,
                     init_pose.to(device=dev, dtype=torch.float64).reshape(-1, 4).contiguous() if init_pose is not None else None]
        x2d, istd, x3d, cam, ur, vr, thr, ini = self.keep
        if out is None:
            self.valid = torch.empty(B, device=dev, dtype=torch.uint8)
            self.pose = torch.empty(B, 4, **f32)
            self.cov = torch.empty(B, 4, 4, **f32)
            self.tr = torch.empty(B, **f32)
        else:                                   # e.g. the typed views of parallel.PackedResults
            self.valid, self.pose, self.cov, self.tr = out.valid, out.pose, out.cov, out.tr
        self.mask = mask if mask is not None else torch.empty(B, P, device=dev, dtype=torch.uint8)
        assert self.mask.shape == (B, P) and self.mask.dtype == torch.uint8 and self.mask.is_contiguous()
        self.diag = torch.empty(B, 4, **f32) if with_diag else None
        self.B = B
        self.args = [x2d.data_ptr(), _strides(x2d), istd.data_ptr(), _strides(istd), x3d.data_ptr(), _strides(x3d),
                     _DTYPES[x2d.dtype], cam.data_ptr(), cam.shape[0], ur.data_ptr(), vr.data_ptr(), ur.shape[0],
                     thr.data_ptr() if thr is not None else None, ini.data_ptr() if ini is not None else None, B, P,
                     float(z_min), float(epnp_istd_thres), int(bool(inlier_opt_only)), int(flags),
                     self.valid.data_ptr(), self.pose.data_ptr(), self.cov.data_ptr(), self.tr.data_ptr(),
                     self.mask.data_ptr(), self.diag.data_ptr() if self.diag is not None else None]

    def run(self, stream=None):
        if self.B == 0:
            return
        st = stream if stream is not None else torch.cuda.current_stream(self.dev).cuda_stream
        if torch.cuda.current_device() != self.dev.index:        # the library launches on the CURRENT HIP device
            with torch.cuda.device(self.dev):
                code = self.lib.mr_pnp_uncert_batched(*self.args, st)
        else:
            code = self.lib.mr_pnp_uncert_batched(*self.args, st)
        if code:
            _lib.check(code)


class PnPEpnpLaunch:
    """A prepared launch of the REFERENCE's flow over device-resident inputs: its initialiser (``mr_epnp_ransac_batched``: the
    launches of csrc/epnp_stages.inc, pnp_uncert_cpu.py:33-68) followed by the LM + covariance launch
    (``mr_pnp_uncert_from_init_batched``) on the same stream.  Outputs, the initialiser's hand-over buffers and its workspace
    (``mr_epnp_workspace_bytes``: 11.5 MB per 1024 objects) are allocated once; ``run()`` only enqueues.  The stages are latency
    chains that leave most issue slots of the chip idle, so several of these launches in flight (``PnPPipeline.submit``)
    overlap almost for free."""

    def __init__(self, coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=0.5, epnp_istd_thres=0.6,
                 epnp_ransac_thres=None, inlier_opt_only=True, flags=0, max_iters=30, with_diag=False, first_round=None,
                 out=None, mask=None, work=None, fused=True, calib=None):
        """calib: None, or (logscale, sd, out) — a float32 device tensor of the four calibration log-scales (read when the launch RUNS), the
        distance-correction constant (0 = none) and a (B,4,4) float32 tensor that receives (s s^T) * cov * (sd / ||t||)^2, written by the LM
        launch's epilogue (fused form only; uncert_prop_pnp_optimizer.py:96-97, monorun_roi_head.py:530-534).
        fused (default): the initialiser stops before its last launch (MR_EPNP_DEFER_REFIT) and the LM launch runs the re-fit's pose
        candidates as its prologue (``mr_pnp_uncert_from_epnp_grouped``: one launch and one pass over the correspondences less; same
        results, bit for bit); False: the two entry points one after the other (``mr_epnp_ransac_batched``, then
        ``mr_pnp_uncert_from_init_batched``).
        out: an object with valid / pose / cov / tr tensors to write the results into (e.g. the typed views of
        parallel.PackedResults: the kernel then writes straight into the buffer a collective sends); mask: the (B,P) uint8 inlier-mask
        buffer; work: a uint8 workspace of at least mr_epnp_workspace_bytes(B, P) bytes that SEVERAL launches may share when they
        only ever run on one stream (stream order keeps that safe).  All three default to tensors of the launch's own."""
        self.lib = lib = _lib.load()
        dev = coords_2d.device
        if dev.type != 'cuda':
            raise RuntimeError('PnPEpnpLaunch needs HIP device tensors (no CPU fallback)')
        self.dev = dev
        B, P = int(coords_2d.shape[0]), int(coords_2d.shape[1])
        assert coords_2d.dtype in _DTYPES and coords_2d_istd.dtype == coords_2d.dtype == coords_3d.dtype
        f32 = dict(device=dev, dtype=torch.float32)
        self.keep = [coords_2d, coords_2d_istd, coords_3d,
                     cam_mats.to(**f32).reshape(-1, 3, 3).contiguous(), u_range.to(**f32).reshape(-1, 2).contiguous(),
                     v_range.to(**f32).reshape(-1, 2).contiguous(),
                     epnp_ransac_thres.to(**f32).reshape(-1).contiguous() if epnp_ransac_thres is not None else None]
        x2d, istd, x3d, cam, ur, vr, thr = self.keep
        self.init_pose = torch.empty(B, 4, device=dev, dtype=torch.float64)
        self.init_mask = torch.empty(B, P, device=dev, dtype=torch.uint8)
        self.init_valid = torch.empty(B, device=dev, dtype=torch.uint8)
        self.init_diag = torch.empty(B, 4, **f32) if with_diag else None
        need = int(lib.mr_epnp_workspace_bytes(B, P)) if B > 0 else 0
        self.work = work if work is not None else torch.empty(need, device=dev, dtype=torch.uint8)
        assert self.work.dtype == torch.uint8 and self.work.numel() >= need and self.work.data_ptr() % 256 == 0
        if out is None:
            self.valid = torch.empty(B, device=dev, dtype=torch.uint8)
            self.pose = torch.empty(B, 4, **f32)
            self.cov = torch.empty(B, 4, 4, **f32)
            self.tr = torch.empty(B, **f32)
        else:
            self.valid, self.pose, self.cov, self.tr = out.valid, out.pose, out.cov, out.tr
        self.mask = mask if mask is not None else torch.empty(B, P, device=dev, dtype=torch.uint8)
        assert self.mask.shape == (B, P) and self.mask.dtype == torch.uint8 and self.mask.is_contiguous()
        self.diag = torch.empty(B, 4, **f32) if with_diag else None
        self.B = B
        head = [x2d.data_ptr(), _strides(x2d), istd.data_ptr(), _strides(istd), x3d.data_ptr(), _strides(x3d), _DTYPES[x2d.dtype], cam.data_ptr(), cam.shape[0]]
        self.args_init = head + [thr.data_ptr() if thr is not None else None, B, P, float(epnp_istd_thres), (int(flags) & 0x1047) | ((max(1, min(30, int(first_round))) << _lib.MR_EPNP_FIRST_ROUND_SHIFT) if first_round is not None else 0),
                                 int(max_iters), self.init_pose.data_ptr(), self.init_mask.data_ptr(), self.init_valid.data_ptr(),
                                 self.init_diag.data_ptr() if self.init_diag is not None else None, None, self.work.data_ptr(), self.work.numel()]
        self.args_lm = head + [ur.data_ptr(), vr.data_ptr(), ur.shape[0], self.init_pose.data_ptr(), self.init_mask.data_ptr(), self.init_valid.data_ptr(),
                               B, P, float(z_min), int(bool(inlier_opt_only)), int(flags), self.valid.data_ptr(), self.pose.data_ptr(), self.cov.data_ptr(),
                               self.tr.data_ptr(), self.mask.data_ptr(), self.diag.data_ptr() if self.diag is not None else None]
        self.calib = calib
        if calib is not None:
            ls, _, co = calib
            if not fused:
                raise ValueError('calib needs the fused form (the LM launch that carries the re-fit writes it)')
            assert ls.dtype == torch.float32 and ls.numel() == 4 and ls.device == dev and co.dtype == torch.float32 and co.is_contiguous() and co.numel() == B * 16
        self.fused = bool(fused)
        if self.fused:
            # the launch set of this one call.  It keeps argument lists (raw pointers into tensors THIS object owns), not this object: no
            # reference cycle, so the workspace / outputs / masks (~11.5 MB per 1024 objects) are freed by refcount when the launch is dropped
            # — a serving loop that builds launches per request does not wait for the cyclic GC
            self._single = PnPEpnpGroupLaunch([self], work=self.work, lm='fused')
            self._single.members = ()

    def run(self, stream=None):
        if self.B == 0:
            return
        if self.fused:
            return self._single.run(stream)
        st = stream if stream is not None else torch.cuda.current_stream(self.dev).cuda_stream
        with torch.cuda.device(self.dev):                 # the library launches on the CURRENT HIP device
            code = self.lib.mr_epnp_ransac_batched(*self.args_init, st)
            if not code:
                code = self.lib.mr_pnp_uncert_from_init_batched(*self.args_lm, st)
        if code:
            _lib.check(code)


class PnPEpnpGroupLaunch:
    """Up to eight prepared ``PnPEpnpLaunch`` objects of the same shape whose initialisers run as ONE launch set
    (``mr_epnp_ransac_grouped``: every launch of csrc/epnp_stages.inc carries the objects of all members), followed by each
    member's own LM + covariance launch, all on the stream ``run`` is given.  Members keep their inputs and outputs; results are
    bit-identical to running them one by one.  Why: HIP runs the launches of at most four streams side by side and the
    initialiser's stages are latency chains that fill a fraction of the chip, so a ``PnPPipeline`` of depth 4 that is fed groups
    of two keeps EIGHT calls' stages in flight (measured on MI355X, reference flow, 1024-object calls: DESIGN.md section 3)."""

    def __init__(self, launches, work=None, lm='fused'):
        """lm: how the members' LM + covariance launches are issued behind the set's initialiser — 'fused' (default): ONE launch over the
        objects of all members that also carries the initialiser's last step, the re-fit's pose candidates (MR_EPNP_DEFER_REFIT +
        ``mr_pnp_uncert_from_epnp_grouped``: one launch and one pass over the correspondences less); 'grouped': ONE launch over the
        objects of all members behind the complete initialiser (``mr_pnp_uncert_from_init_grouped``: the set pays its slowest object
        once); 'side_by_side': one launch per
        member, the second and later ones with MR_ANY_ORDER; 'serial': one per member in stream order.  Same results.
        work: a uint8 workspace of at least mr_epnp_workspace_bytes(len(launches) * B, P) bytes (shared between groups that only
        ever run on one stream), or None to allocate one."""
        self.members = list(launches)
        n = len(self.members)
        if not 1 <= n <= 8:
            raise ValueError('PnPEpnpGroupLaunch takes 1 to 8 launches')
        f = self.members[0]
        self.lib, self.dev, self.B = f.lib, f.dev, f.B
        ai = f.args_init
        same = lambda m: (m.B == f.B and m.dev == f.dev and tuple(m.args_init[1]) == tuple(ai[1]) and tuple(m.args_init[3]) == tuple(ai[3]) and
                          tuple(m.args_init[5]) == tuple(ai[5]) and m.args_init[6] == ai[6] and m.args_init[8] == ai[8] and m.args_init[11:15] == ai[11:15] and
                          (m.args_init[9] is None) == (ai[9] is None) and (m.args_init[18] is None) == (ai[18] is None))
        if not all(same(m) for m in self.members):
            raise ValueError('the members of a group must share shape, strides, dtype, camera batching, thresholds and flags')
        arr = lambda k: (ctypes.c_void_p * n)(*[m.args_init[k] for m in self.members])
        self._arrays = [arr(k) for k in (0, 2, 4, 7, 9, 15, 16, 17, 18)]
        x2d, istd, x3d, cam, thr, ipose, imask, ivalid, idiag = self._arrays
        P = ai[11]
        need = int(self.lib.mr_epnp_workspace_bytes(n * f.B, P)) if f.B > 0 else 0
        self.work = work if work is not None else torch.empty(need, device=f.dev, dtype=torch.uint8)
        assert self.work.dtype == torch.uint8 and self.work.numel() >= need and self.work.data_ptr() % 256 == 0
        if lm not in ('fused', 'grouped', 'side_by_side', 'serial'):
            raise ValueError("lm must be 'fused', 'grouped', 'side_by_side' or 'serial'")
        self.args = [n, x2d, ai[1], istd, ai[3], x3d, ai[5], ai[6], cam, ai[8], thr, f.B, P, ai[12],
                     int(ai[13]) | (_lib.MR_EPNP_DEFER_REFIT if lm == 'fused' else 0), ai[14],
                     ipose, imask, ivalid, idiag, self.work.data_ptr(), self.work.numel()]
        al = f.args_lm
        lm_same = lambda m: (m.args_lm[11] == al[11] and m.args_lm[17:20] == al[17:20] and (m.args_lm[25] is None) == (al[25] is None))
        if not all(lm_same(m) for m in self.members):
            raise ValueError('the members of a group must share range batching, z_min, inlier_opt_only and flags')
        self.lm = lm
        larr = lambda k: (ctypes.c_void_p * n)(*[m.args_lm[k] for m in self.members])
        self._lm_arrays = {k: larr(k) for k in (9, 10, 12, 13, 14, 20, 21, 22, 23, 24, 25)}
        la = self._lm_arrays
        self.args_lm = [n, x2d, al[1], istd, al[3], x3d, al[5], al[6], cam, al[8], la[9], la[10], al[11], la[12], la[13], la[14], f.B, P, al[17], al[18], al[19],
                        la[20], la[21], la[22], la[23], la[24], la[25]]
        cal = [getattr(m, 'calib', None) for m in self.members]
        if any(c is not None for c in cal) and (any(c is None for c in cal) or any(c[0].data_ptr() != cal[0][0].data_ptr() or c[1] != cal[0][1] for c in cal)):
            raise ValueError('the members of a group must share the calibration (log-scale tensor and distance constant), or have none')
        if cal[0] is not None and lm != 'fused':
            raise ValueError("calibrated covariances are written by the 'fused' form")
        self._calib_arr = (ctypes.c_void_p * n)(*[c[2].data_ptr() for c in cal]) if cal[0] is not None else None
        self.args_fused = self.args_lm[:16] + [idiag] + self.args_lm[16:] + [cal[0][0].data_ptr() if cal[0] is not None else None, float(cal[0][1]) if cal[0] is not None else 0.0,
                                                                             self._calib_arr, self.work.data_ptr(), self.work.numel()]
        self._lm_any = []
        for m in self.members:                             # args_lm with MR_ANY_ORDER in its flags (argument 19 of mr_pnp_uncert_from_init_batched)
            a = list(m.args_lm)
            a[19] = int(a[19]) | _lib.MR_ANY_ORDER
            self._lm_any.append(a)

    def run(self, stream=None):
        if self.B == 0:
            return
        st = stream if stream is not None else torch.cuda.current_stream(self.dev).cuda_stream
        with torch.cuda.device(self.dev):
            code = self.lib.mr_epnp_ransac_grouped(*self.args, st)
            if self.lm == 'fused':
                if not code:
                    code = self.lib.mr_pnp_uncert_from_epnp_grouped(*self.args_fused, st)
                if code:
                    _lib.check(code)
                return
            if not code and self.lm == 'grouped' and len(self.members) > 1:
                code = self.lib.mr_pnp_uncert_from_init_grouped(*self.args_lm, st)
                if code:
                    _lib.check(code)
                return
            for k, m in enumerate(self.members):
                if code:
                    break
                # the first call's LM launch waits for the set's initialiser launches (stream order); the others carry MR_ANY_ORDER:
                # they start as soon as the one in front of them has started, i.e. the set's LM launches run side by side instead
                # of each waiting for the slowest object of the one before
                code = self.lib.mr_pnp_uncert_from_init_batched(*(m.args_lm if (k == 0 or self.lm != 'side_by_side') else self._lm_any[k]), st)
        if code:
            _lib.check(code)


class PnPPipeline:
    """Several prepared launches in flight: ``submit`` issues them round-robin on ``depth`` internal HIP streams.

    Why: one launch of B = 1024 objects lasts as long as its slowest object (an object with 17 LM iterations keeps a handful
    of SIMDs busy for ~50 us while the other 1000 objects finished after ~35 us), and launches on ONE stream serialise, so a
    single stream pays that tail on every step.  Launches on different streams overlap: the next batches fill the SIMDs the
    tail leaves idle.  Every launch is still one full fused kernel over its own batch with its own output buffers; results
    are bit-identical to the single-stream ones (objects are independent; nothing is shared between launches).

        pipe = PnPPipeline(device, depth=4)
        ev = pipe.submit(launch)            # enqueue; `ev` completes when launch.valid / pose / cov / mask are written
        torch.cuda.current_stream().wait_event(ev)      # where (and when) a consumer stream needs them
        pipe.drain()                        # or: host-wait for everything submitted so far

    A launch object owns its output buffers, so the SAME PnPLaunch must not be submitted again before its previous run has
    completed unless it lands on the same internal stream (``slot`` pins a launch to a stream: give each buffer set a fixed
    slot and re-submissions are stream-ordered).  ``after`` = an event the inputs depend on (produced on another stream)."""

    def __init__(self, device, depth=4, record_events=True, avoid=(), verify=True):
        """depth: launches in flight asked for.  HIP maps a process's streams onto a few hardware queues (4 on this stack, the
        default stream's included) and two streams on one queue serialise — which of torch's pool streams collide is an internal
        of the runtime (on MI355X / ROCm 7.2 the first four pool streams land on three queues: 23 instead of 34 M solves/s).
        With verify=True the constructor therefore MEASURES it: candidates are taken from the pool and kept only if a 40 us
        one-wave spin kernel (mr_spin) on them runs side by side with one on every stream already kept — and on every stream in
        `avoid` (e.g. the side stream of parallel.RcclAllGather).  ``self.depth`` is the number of streams found (<= depth)."""
        self.dev = torch.device(device)
        if self.dev.type != 'cuda':
            raise RuntimeError('PnPPipeline needs a HIP device (no CPU fallback)')
        want = max(1, int(depth))
        self.overlap_test = None
        if verify and want > 1:
            self.streams, self.overlap_test = self._pick_streams(want, list(avoid))
        else:
            self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(want)]
        self.depth = len(self.streams)
        self.handles = [s.cuda_stream for s in self.streams]
        self.record_events = record_events
        self._next = 0

    def _pick_streams(self, want, avoid):
        import time
        lib = _lib.load()
        US = 40

        def wall(a, b):
            best = 1e9
            for _ in range(3):
                a.synchronize(); b.synchronize()
                t0 = time.perf_counter()
                _lib.check(lib.mr_spin(US, a.cuda_stream))
                _lib.check(lib.mr_spin(US, b.cuda_stream))
                a.synchronize(); b.synchronize()
                best = min(best, time.perf_counter() - t0)
            return best * 1e6
        with torch.cuda.device(self.dev):
            torch.cuda.synchronize(self.dev)
            first = torch.cuda.Stream(device=self.dev)
            serial = wall(first, first)                      # two spins on ONE stream: what a collision looks like
            kept, tested = [first], 0
            if any(wall(first, a) > serial - 0.5 * US for a in avoid):
                kept = []
            for _ in range(8 * want + 8):
                if len(kept) >= want:
                    break
                c = torch.cuda.Stream(device=self.dev)
                tested += 1
                if any(c.cuda_stream == k.cuda_stream for k in kept):
                    continue
                if all(wall(c, k) < serial - 0.5 * US for k in kept + avoid):      # side by side: about one spin shorter than in series
                    kept.append(c)
            if not kept:
                kept = [first]
            torch.cuda.synchronize(self.dev)
        return kept, {'asked': want, 'found': len(kept), 'candidates_tested': tested + 1, 'serial_pair_us': serial}

    def flags_for(self, B, P=784):
        """The MR_WAVES bits for launches of B objects x P points issued through this pipeline.  The library chooses the number of
        wavefronts per object from ONE launch's size (4 up to B = 2048, 2 beyond: more waves shorten an object's latency chain,
        fewer cost fewer instructions per object); with `depth` launches in flight the chip holds depth x B objects, so the same
        rule is applied to that number (measured on MI355X, 1024-object launches, depth 4: 34 instead of 30 M solves/s)."""
        with torch.cuda.device(self.dev):
            w = int(_lib.load().mr_pick_waves(int(B) * self.depth, int(P)))       # the library's own rule (device properties included)
        if w < 0:
            _lib.check(w)
        return w << _lib.MR_WAVES_SHIFT

    def submit(self, launch, slot=None, after=None):
        k = (self._next if slot is None else int(slot)) % self.depth
        self._next += 1
        s = self.streams[k]
        if after is not None:
            s.wait_event(after)
        launch.run(self.handles[k])
        if not self.record_events:
            return None
        ev = getattr(launch, '_done_event', None)
        if ev is None:                                # created once per launch object (per output-buffer set)
            ev = launch._done_event = torch.cuda.Event()
        ev.record(s)
        return ev

    def drain(self):
        for s in self.streams:
            s.synchronize()


def pnp6_refine_device(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, inlier_mask_u8, pose4, valid4_u8, z_min=0.5,
                       flags=0, with_diag=False):
    """Second launch of the 6-DoF mode (``mr_pnp6_refine_batched``): 6-DoF LM from the 4-DoF result on its inlier set.
    Returns (valid u8 (B,), pose6 f32 (B,6) [rx,ry,rz,tx,ty,tz], cov6 f32 (B,6,6), diag f32 (B,2)|None)."""
    lib = _lib.load()
    dev = coords_2d.device
    B, P = int(coords_2d.shape[0]), int(coords_2d.shape[1])
    dt = coords_2d.dtype if coords_2d.dtype in _DTYPES else torch.float32
    prep = lambda t: t.detach() if (t.dtype == dt and t.device == dev) else t.detach().to(device=dev, dtype=dt)
    x2d, istd, x3d = prep(coords_2d), prep(coords_2d_istd), prep(coords_3d)
    f32 = dict(device=dev, dtype=torch.float32)
    cam = cam_mats.detach().to(**f32).reshape(-1, 3, 3).contiguous()
    ur = u_range.detach().to(**f32).reshape(-1, 2).contiguous()
    vr = v_range.detach().to(**f32).reshape(-1, 2).contiguous()
    valid = torch.empty(B, device=dev, dtype=torch.uint8)
    pose6 = torch.empty(B, 6, **f32)
    cov6 = torch.empty(B, 6, 6, **f32)
    diag = torch.empty(B, 2, **f32) if with_diag else None
    if B > 0:
        with torch.cuda.device(dev):
            _lib.check(lib.mr_pnp6_refine_batched(
                x2d.data_ptr(), _strides(x2d), istd.data_ptr(), _strides(istd), x3d.data_ptr(), _strides(x3d), _DTYPES[dt],
                cam.data_ptr(), cam.shape[0], ur.data_ptr(), vr.data_ptr(), ur.shape[0],
                inlier_mask_u8.contiguous().data_ptr(), pose4.contiguous().data_ptr(), valid4_u8.contiguous().data_ptr(), B, P, float(z_min), int(flags),
                valid.data_ptr(), pose6.data_ptr(), cov6.data_ptr(), diag.data_ptr() if diag is not None else None,
                torch.cuda.current_stream(dev).cuda_stream))
    return valid, pose6, cov6, diag


def exact_hessian_device(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, pose4, inlier_mask_u8, valid_u8, z_min=0.5,
                         with_hessian=False):
    """`forward_exact_hessian=True` on device tensors (hessian.py:5-64 + pnp_uncert.py:63-85 of the reference): the exact Hessian of
    the masked cost at pose4 (B,4) f32 and its inverse.  inlier_mask_u8 (B,P) u8 | None, valid_u8 (B,) u8 = the 4-DoF solve's flags.
    Returns (valid u8 (B,), cov f32 (B,4,4), hess f32 (B,4,4) | None)."""
    lib = _lib.load()
    dev = coords_2d.device
    B, P = int(coords_2d.shape[0]), int(coords_2d.shape[1])
    dt = coords_2d.dtype if coords_2d.dtype in _DTYPES else torch.float32
    prep = lambda t: t.detach() if (t.dtype == dt and t.device == dev) else t.detach().to(device=dev, dtype=dt)
    x2d, istd, x3d = prep(coords_2d), prep(coords_2d_istd), prep(coords_3d)
    f32 = dict(device=dev, dtype=torch.float32)
    cam = cam_mats.detach().to(**f32).reshape(-1, 3, 3).contiguous()
    ur = u_range.detach().to(**f32).reshape(-1, 2).contiguous()
    vr = v_range.detach().to(**f32).reshape(-1, 2).contiguous()
    valid = valid_u8.detach().to(device=dev, dtype=torch.uint8).clone().contiguous()
    pose = pose4.detach().to(**f32).contiguous()
    mask = inlier_mask_u8.detach().to(device=dev, dtype=torch.uint8).contiguous() if inlier_mask_u8 is not None else None
    cov = torch.empty(B, 4, 4, **f32)
    hess = torch.empty(B, 4, 4, **f32) if with_hessian else None
    if B > 0:
        with torch.cuda.device(dev):
            _lib.check(lib.mr_pnp_exact_hessian_batched(
                x2d.data_ptr(), _strides(x2d), istd.data_ptr(), _strides(istd), x3d.data_ptr(), _strides(x3d), _DTYPES[dt],
                cam.data_ptr(), cam.shape[0], ur.data_ptr(), vr.data_ptr(), ur.shape[0],
                pose.data_ptr(), mask.data_ptr() if mask is not None else None, B, P, float(z_min),
                valid.data_ptr(), hess.data_ptr() if hess is not None else None, cov.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    return valid, cov, hess


def pnp_uncert(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=0.5, epnp_istd_thres=1.0,
               epnp_ransac_thres=None, inlier_opt_only=False, forward_exact_hessian=False, use_6dof=False, initialiser=None, cov_symeig_rule=False,
               epnp_first_round=None):
    """Functional form of the op on torch tensors (argument names and defaults: pnp_uncert.py:7-11 of the reference).

    coords_2d / coords_2d_istd (B,P,2), coords_3d (B,P,3), cam_mats (B|1,3,3), u_range / v_range (B|1,2),
    epnp_ransac_thres (B,) or None — any device; host tensors are staged through the GPU.
    forward_exact_hessian=True (no shipped config, e.g. configs/kitti_car.py:123 sets False; the reference's own exact Hessian no
    longer runs on torch >= 2): pose_cov = inverse of the exact Hessian of the masked cost (hessian.py:5-64) instead of
    inverse(J^T J) — a second launch (exact_hessian_device); ignored together with use_6dof=True (the 6-DoF covariance is the
    solver's J^T J).
    initialiser (not a reference keyword; default = DEFAULT_INITIALISER = 'epnp'): 'epnp' = the reference's own initialiser —
    cv2.solvePnPRansac(..., iterationsCount=30, flags=SOLVEPNP_EPNP), pnp_uncert_cpu.py:33-68 — as its own sequence of launches in front
    of the LM launch (the published algorithm as DESIGN.md §5 restates it: inlier sets, start pose and hence the returned pose are the
    reference flow's, up to what OpenCV's own build would do); 'k0' = the explicit FAST MODE: this repository's deterministic consensus
    initialiser inside the fused kernel (one launch, ~5 x the throughput, inlier sets that differ from the reference flow's on ~13 % of
    the objects).
    epnp_first_round (with initialiser='epnp'; not a reference keyword): how many of the 30 speculative RANSAC hypotheses are solved for
    every object before the replayed loop is consulted (default: the library's rule — 10 for calls or launch sets of fewer than 2048 objects,
    3 beyond —; the rest only where the loop wants them; 30 = one round, the setting
    for outlier-heavy candidate sets one call at a time).  Never changes a result.
    cov_symeig_rule (not a reference keyword): also apply, per object, the eigenvalue test of the reference's fallback branch
    (pnp_uncert.py:77-85: keep an object only if lambda_min(h) > max(1e-6 lambda_max(h), 0), else ret_val = False and pose_cov = I).
    Default False = this kernel's own rule (an object is dropped when h has no Cholesky factorisation), which is what the
    reference does whenever torch.inverse does not raise.
    use_6dof=False (every shipped config): returns (ret_val (B,) bool, r_vec (B,1) yaw, t_vec (B,3), pose_cov (B,4,4) covariance of
    [yaw, t], inlier_mask (B,P) bool) on the device and in the dtype of coords_2d — the reference's tuple.
    use_6dof=True: the flag the reference declares and never reads (pnp_uncert.py:11) made real — after the 4-DoF solve (mask,
    initial pose) a second launch refines all six pose parameters with the same residual and LM: r_vec becomes the (B,3)
    angle-axis vector, pose_cov the (B,6,6) covariance of [rx, ry, rz, tx, ty, tz] (solver Jacobian); ret_val additionally
    requires the 6-DoF solve to be usable.  Because the flag is dead in the reference, every shipped config keeps running
    the 4-DoF path.
    """
    with torch.no_grad():
        src_dev = coords_2d.device
        if src_dev.type != 'cuda':
            if not torch.cuda.is_available():
                raise RuntimeError('monorun_amd.ops.pnp_uncert needs an MI355X (HIP) device; no CPU fallback exists')
            dev = torch.device('cuda', torch.cuda.current_device())
            mv = lambda t: t.to(dev) if t is not None else None
            coords_2d, coords_2d_istd, coords_3d = mv(coords_2d), mv(coords_2d_istd), mv(coords_3d)
        if initialiser is None:
            initialiser = DEFAULT_INITIALISER
        if initialiser == 'epnp':
            valid, pose, cov, _, mask = pnp_uncert_epnp_device(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=z_min,
                                                               epnp_istd_thres=epnp_istd_thres, epnp_ransac_thres=epnp_ransac_thres,
                                                               inlier_opt_only=inlier_opt_only, first_round=epnp_first_round)[:5]
        elif initialiser == 'k0':
            valid, pose, cov, _, mask, _ = pnp_uncert_device(
                coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, z_min=z_min,
                epnp_istd_thres=epnp_istd_thres, epnp_ransac_thres=epnp_ransac_thres, inlier_opt_only=inlier_opt_only)
        else:
            raise ValueError(f"initialiser must be 'k0' or 'epnp', got {initialiser!r}")
        odt = coords_2d.dtype
        if use_6dof and cov_symeig_rule:
            raise ValueError('cov_symeig_rule tests the 4x4 covariance of [yaw, t] (pnp_uncert.py:77-85); it has no meaning for the 6x6 '
                             'covariance that use_6dof=True returns')
        if use_6dof:
            valid6, pose6, cov6, _ = pnp6_refine_device(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, mask, pose, valid, z_min=z_min)
            return ((valid & valid6).to(device=src_dev, dtype=torch.bool), pose6[:, :3].to(device=src_dev, dtype=odt),
                    pose6[:, 3:].to(device=src_dev, dtype=odt), cov6.to(device=src_dev, dtype=odt), mask.to(device=src_dev, dtype=torch.bool))
        if forward_exact_hessian:
            valid, cov, _ = exact_hessian_device(coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, pose, mask, valid, z_min=z_min)
        if cov_symeig_rule:
            cov = cov.contiguous()
            cov_symeig_rule_device(valid, cov)
        ret_val = valid.to(device=src_dev, dtype=torch.bool)
        r_vec = pose[:, :1].to(device=src_dev, dtype=odt)
        t_vec = pose[:, 1:].to(device=src_dev, dtype=odt)
        pose_cov = cov.to(device=src_dev, dtype=odt)
        inlier_mask = mask.to(device=src_dev, dtype=torch.bool)
    return ret_val, r_vec, t_vec, pose_cov, inlier_mask


@PNP.register_module()
class PnPUncert(torch.nn.Module):

    def __init__(self, z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, coord_istd_normalize=False,
                 forward_exact_hessian=False, use_6dof=False, eps=1e-6, initialiser=None, epnp_first_round=None, cov_symeig_rule=False):
        """Module form (constructor keywords of the reference, pnp_uncert.py:93-99; no parameters, no buffers).
        epnp_istd_thres: a point is an istd inlier when both of its istd components reach this factor times the object's
        mean; inlier_opt_only: the LM refines on the inlier set only; coord_istd_normalize: divide the istd map by its
        per-object mean (clamped at eps) first.  initialiser ('k0' | 'epnp'), epnp_first_round, cov_symeig_rule (not reference
        keywords): see ``pnp_uncert``.  The reference's own config dict builds the REFERENCE's flow (initialiser='epnp', the default since
        round 5); ``initialiser='k0'`` selects the one-launch fast mode."""
        super().__init__()
        if initialiser is None:
            initialiser = DEFAULT_INITIALISER
        if initialiser not in ('k0', 'epnp'):
            raise ValueError(f"initialiser must be 'k0' or 'epnp', got {initialiser!r}")
        if cov_symeig_rule and use_6dof:
            raise ValueError('cov_symeig_rule applies to the 4-DoF covariance only (use_6dof=True returns a 6x6 one)')
        self.initialiser, self.epnp_first_round, self.cov_symeig_rule = initialiser, epnp_first_round, cov_symeig_rule
        self.z_min, self.epnp_istd_thres, self.inlier_opt_only = z_min, epnp_istd_thres, inlier_opt_only
        self.coord_istd_normalize, self.eps = coord_istd_normalize, eps
        self.forward_exact_hessian, self.use_6dof = forward_exact_hessian, use_6dof

    def forward(self, coords_2d, coords_2d_istd, coords_3d, cam_mats, u_range, v_range, epnp_ransac_thres=None):
        istd = coords_2d_istd
        if self.coord_istd_normalize:
            istd = istd / istd.mean(dim=(1, 2), keepdim=True).clamp(min=self.eps)
        return pnp_uncert(coords_2d, istd, coords_3d, cam_mats, u_range, v_range, z_min=self.z_min,
                          epnp_istd_thres=self.epnp_istd_thres, epnp_ransac_thres=epnp_ransac_thres,
                          inlier_opt_only=self.inlier_opt_only, forward_exact_hessian=self.forward_exact_hessian,
                          use_6dof=self.use_6dof, initialiser=self.initialiser, epnp_first_round=self.epnp_first_round,
                          cov_symeig_rule=self.cov_symeig_rule)

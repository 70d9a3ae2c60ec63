"""ctypes loader / builder for libmonorun_pnp.so (the C ABI declared in include/monorun_pnp.h).

The library is built in-tree by hipcc for gfx950 (``build()``; also driven by
``__graft_entry__.build()``) and loaded with ctypes — cffi, which the reference uses
(/root/reference/monorun/ops/least_squares/setup.py:12-24), is not assumed to exist.
There is NO CPU fallback: if the library is missing or no HIP device is present the ops raise.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, 'csrc', 'monorun_pnp.hip')
INCLUDE = os.path.join(_ROOT, 'include')
SO = os.environ.get('MR_PNP_SO') or os.path.join(_HERE, 'libmonorun_pnp.so')     # MR_PNP_SO: A/B-test another build of the library

MR_F32, MR_F16, MR_F64, MR_BF16 = 0, 1, 2, 3
MR_MEAN_AUTO, MR_MEAN_SEQUENTIAL, MR_MEAN_PAIRWISE = 0, 1, 2
MR_NO_ISTD_MASK, MR_COV_NONE, MR_COV_CERES, MR_ANY_ORDER = 0x4, 0x8, 0x10, 0x20
MR_EPNP_REFIT_F32 = 0x40
MR_EPNP_DEFER_REFIT = 0x80
MR_EPNP_CV_EARLY_RETURN = 0x1000
MR_WAVES_SHIFT = 8
MR_LM_MAXIT_SHIFT = 16
MR_EPNP_FIRST_ROUND_SHIFT = 24

HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared']

_lib = None


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [SRC, os.path.join(_HERE, "csrc", "pnp_kernel.inc"), os.path.join(_HERE, "csrc", "pnp_kernel_body.inc"), os.path.join(_HERE, "csrc", "pnp6_kernel.inc"), os.path.join(_HERE, "csrc", "hessian_kernel.inc"), os.path.join(_HERE, "csrc", "pnp_noc_kernel.inc"), os.path.join(_HERE, "csrc", "epnp_kernel.inc"), os.path.join(_HERE, "csrc", "epnp_eig_low4.inc"), os.path.join(_HERE, "csrc", "epnp_stages.inc"), os.path.join(_HERE, "csrc", "epnp_consensus_body.inc"), os.path.join(_HERE, "csrc", "kitti_eval_kernel.inc"), os.path.join(INCLUDE, "monorun_pnp.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    if not force and not _stale():
        return SO
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    extra = os.environ.get('MR_HIPCC_EXTRA', '').split()
    cmd = [hipcc] + HIPCC_FLAGS + extra + ['-I', INCLUDE, SRC, '-o', SO]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return SO


def load():
    """Return the ctypes handle; raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise RuntimeError(
            f'{SO} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950). monorun_amd has no CPU fallback.')
    lib = ctypes.CDLL(SO)
    vp, i32, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    i64p = ctypes.POINTER(ctypes.c_int64)
    lib.mr_pnp_version.restype = i32
    lib.mr_pnp_error_string.restype = ctypes.c_char_p
    lib.mr_pnp_error_string.argtypes = [i32]
    lib.mr_pnp_last_hip_error.restype = i32
    lib.mr_pnp_device_count.restype = i32
    lib.mr_spin.restype = i32
    lib.mr_pick_waves.restype = i32
    lib.mr_pick_waves.argtypes = [i32, i32]
    lib.mr_spin.argtypes = [i32, vp]
    lib.mr_pnp_uncert_batched.restype = i32
    lib.mr_pnp_uncert_batched.argtypes = [
        vp, i64p, vp, i64p, vp, i64p, i32,          # x2d, istd, x3d (+strides), in_dtype
        vp, i32, vp, vp, i32,                       # cam_mats, cam_batch, u_range, v_range, range_batch
        vp, vp, i32, i32,                           # ransac_thr, init_pose, B, P
        f32, f32, i32, i32,                         # z_min, istd_thres, inlier_opt_only, flags
        vp, vp, vp, vp, vp, vp, vp]                 # valid, pose, cov, tr, mask, diag, stream
    lib.mr_epnp_ransac_batched.restype = i32
    lib.mr_epnp_ransac_batched.argtypes = [vp, i64p, vp, i64p, vp, i64p, i32, vp, i32, vp, i32, i32, f32, i32, i32, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.mr_epnp_ransac_grouped.restype = i32
    lib.mr_epnp_ransac_grouped.argtypes = [i32, vp, i64p, vp, i64p, vp, i64p, i32, vp, i32, vp, i32, i32, f32, i32, i32, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.mr_epnp_workspace_bytes.restype = ctypes.c_size_t
    lib.mr_epnp_workspace_bytes.argtypes = [i32, i32]
    lib.mr_pnp_uncert_from_init_batched.restype = i32
    lib.mr_pnp_uncert_from_init_batched.argtypes = [vp, i64p, vp, i64p, vp, i64p, i32, vp, i32, vp, vp, i32, vp, vp, vp, i32, i32, f32, i32, i32,
                                                    vp, vp, vp, vp, vp, vp, vp]
    lib.mr_pnp_uncert_from_init_grouped.restype = i32
    lib.mr_pnp_uncert_from_init_grouped.argtypes = [i32, vp, i64p, vp, i64p, vp, i64p, i32, vp, i32, vp, vp, i32, vp, vp, vp, i32, i32, f32, i32, i32,
                                                    vp, vp, vp, vp, vp, vp, vp]
    lib.mr_pnp_uncert_from_epnp_grouped.restype = i32
    lib.mr_pnp_uncert_from_epnp_grouped.argtypes = [i32, vp, i64p, vp, i64p, vp, i64p, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, f32, i32, i32,
                                                    vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, ctypes.c_size_t, vp]
    lib.mr_cov_symeig_rule.restype = i32
    lib.mr_cov_symeig_rule.argtypes = [vp, vp, i32, vp, vp]
    lib.mr_pnp_exact_hessian_batched.restype = i32
    lib.mr_pnp_exact_hessian_batched.argtypes = [vp, i64p, vp, i64p, vp, i64p, i32, vp, i32, vp, vp, i32, vp, vp, i32, i32, f32, vp, vp, vp, vp]
    lib.mr_pnp6_refine_batched.restype = i32
    lib.mr_pnp6_refine_batched.argtypes = [vp, i64p, vp, i64p, vp, i64p, i32, vp, i32, vp, vp, i32, vp, vp, vp, i32, i32, f32, i32, vp, vp, vp, vp, vp]
    dp = ctypes.POINTER(ctypes.c_double)
    lib.pnp_uncert.restype = None
    lib.pnp_uncert.argtypes = [dp, dp, dp, dp, dp, ctypes.POINTER(i32), dp, dp, dp, i32, dp]
    if hasattr(lib, 'mr_noc_decode_batched'):
        lib.mr_noc_decode_batched.restype = i32
        lib.mr_noc_decode_batched.argtypes = [
            vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32,
            vp, vp, vp, vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, f32, f32,
            vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.mr_pnp_from_head_batched.restype = i32
    lib.mr_pnp_from_head_batched.argtypes = [
        vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32,
        vp, vp, vp, vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, f32, f32,
        vp, i32, vp, vp, i32, f32, f32, i32, i32,
        vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, f32, vp, vp]
    lib.mr_roi_align_avg.restype = i32
    lib.mr_roi_align_avg.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp, vp]
    for name in ('pnp_noc_uncert', 'pnp_noc_cov_uncert'):
        f = getattr(lib, name)
        f.restype = None
        f.argtypes = [dp, dp, dp, dp, dp, dp, dp, ctypes.POINTER(i32), dp, i32, dp, ctypes.c_double]
    lib.mr_pnp_noc_batched.restype = i32
    lib.mr_pnp_noc_batched.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, ctypes.c_double, i32, i32, vp, vp, vp, vp]
    lib.mr_nms_bev_batched.restype = i32
    lib.mr_nms_bev_batched.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp]
    i64 = ctypes.c_int64
    lib.mr_kitti_overlaps.restype = i32
    lib.mr_kitti_overlaps.argtypes = [i32, i32, i32, i32, vp, vp, vp, i64, vp, vp, vp, vp]
    lib.mr_kitti_match_workspace_bytes.restype = i64
    lib.mr_kitti_match_workspace_bytes.argtypes = [i32, i32]
    lib.mr_kitti_match.restype = i32
    lib.mr_kitti_match.argtypes = [i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i64, i64,
                                   vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    _lib = lib
    return lib


def check(code):
    if code != 0:
        lib = load()
        raise RuntimeError(f'libmonorun_pnp: {lib.mr_pnp_error_string(code).decode()} '
                           f'(code {code}, hip error {lib.mr_pnp_last_hip_error()})')


EXPORTED_SYMBOLS = ('mr_pnp_version', 'mr_spin', 'mr_pick_waves', 'mr_pnp_error_string', 'mr_pnp_last_hip_error', 'mr_pnp_device_count',
                    'mr_pnp_uncert_batched', 'mr_epnp_ransac_batched', 'mr_epnp_ransac_grouped', 'mr_epnp_workspace_bytes', 'mr_pnp_uncert_from_init_batched', 'mr_pnp_uncert_from_init_grouped', 'mr_pnp_uncert_from_epnp_grouped', 'mr_cov_symeig_rule', 'mr_pnp6_refine_batched', 'mr_pnp_exact_hessian_batched', 'pnp_uncert', 'mr_noc_decode_batched', 'mr_pnp_from_head_batched', 'mr_nms_bev_batched', 'pnp_noc_uncert', 'pnp_noc_cov_uncert', 'mr_pnp_noc_batched',
                    'mr_kitti_overlaps', 'mr_kitti_match_workspace_bytes', 'mr_kitti_match', 'mr_roi_align_avg')

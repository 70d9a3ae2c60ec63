/*
 * monorun_pnp.h — C ABI of libmonorun_pnp.so (MI355X / gfx950 HIP implementation of MonoRUn's
 * uncertainty-aware 4-DoF PnP hot path).  Plain pointers and sizes only; no torch / C++ types.
 *
 * Reference interfaces replaced (paths relative to the MonoRUn tree, /root/reference):
 *   - monorun/ops/least_squares/src/ext.h:1-13 .......... `pnp_uncert` (per-object, host fp64 buffers;
 *     bound by cffi in monorun/ops/least_squares/setup.py:12-24, called at pnp_uncert_cpu.py:102-106)
 *   - monorun/ops/least_squares/pnp_uncert_cpu.py:128-209  `u2d_pnp_cpu` (batch driver: istd mask,
 *     EPnP/RANSAC initialiser, per-object LM)  +  monorun/ops/least_squares/pnp_uncert.py:60-85
 *     (approx Hessian + inverse)  ->  `mr_pnp_uncert_batched` (one fused kernel, device pointers)
 *   - the elementwise decode chain in front of the PnP (fcn_noc_decoder.py:225-267, noc_coder.py:50-73,
 *     multiclass_norm_dim_coder.py:28-36, distance_invar_proj_error_coder.py:39-60,
 *     uncert_prop_pnp_optimizer.py:73-88, roi_align of coord_2d at monorun_roi_head.py:521-523)
 *     -> `mr_noc_decode_batched`
 *
 * Conventions: every function returns 0 on success or a negative MR_ERR_* code; nothing is allocated
 * that the caller must free; device entry points are asynchronous on the given HIP stream and touch
 * only caller-owned memory.  Numerical failure of a solve is reported per object in `valid`,
 * never as an error code (pnp_uncert_cpu.py:119-125).
 *
 * POINTER CONVENTION.  In every `mr_*` entry point EVERY data pointer — tensors, per-object vectors, the small
 * constant tables (cam_mats, u_range, v_range, dim_means, dim_stds, noc_means, noc_stds, cov_calib_logscale, offsets,
 * thresholds ...) and all outputs — is a DEVICE pointer (hipMalloc / torch CUDA tensor memory), dereferenced only by
 * the kernels, in stream order.  The only host pointers are the three `*_strides` arrays of mr_pnp_uncert_batched
 * (3 x int64, read before the call returns) and `stream`.  The three reference entry points `pnp_uncert`,
 * `pnp_noc_uncert`, `pnp_noc_cov_uncert` keep the reference's convention instead: HOST fp64 buffers, blocking.
 * Multi-device: a call runs on the CURRENT HIP device (hipSetDevice / torch.cuda.device); per-device state inside the
 * library (LDS opt-in, staging buffers of the host entry points) is keyed by device.
 */
#ifndef MONORUN_PNP_H_
#define MONORUN_PNP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MR_PNP_VERSION 100            /* 0.1.0 */

/* input element types of the correspondence tensors */
#define MR_F32 0
#define MR_F16 1
#define MR_F64 2
#define MR_BF16 3                      /* head outputs only (mr_noc_decode_batched, mr_pnp_from_head_batched) */

/* error codes */
#define MR_OK                 0
#define MR_ERR_BAD_ARGUMENT  (-1)
#define MR_ERR_UNSUPPORTED   (-2)     /* P too large for LDS, unknown dtype, ... */
#define MR_ERR_HIP           (-3)     /* a HIP runtime call failed; see mr_pnp_last_hip_error() */
#define MR_ERR_NO_DEVICE     (-4)

/* flags for mr_pnp_uncert_batched */
#define MR_MEAN_AUTO          0x0     /* pick numpy's order from the istd strides (see below) */
#define MR_MEAN_SEQUENTIAL    0x1     /* numpy order for a C-contiguous (B,P,2) float32 array */
#define MR_MEAN_PAIRWISE      0x2     /* numpy order when the point axis is the contiguous one */
#define MR_MEAN_MASK          0x3
#define MR_NO_ISTD_MASK       0x4     /* skip the istd inlier test: every point is a candidate */
#define MR_COV_NONE           0x8     /* do not compute the pose covariance (cov left untouched) */
#define MR_COV_CERES          0x10    /* covariance with the solver's (Ceres/autodiff) Jacobian instead of the
                                         torch Jacobian of jacobian.py (what `pnp_uncert`'s result_cov is) */
#define MR_ANY_ORDER          0x20    /* mr_pnp_uncert_batched and mr_pnp_uncert_from_init_batched (every other entry point ignores the bit): the
                                         launch may start before earlier work on the same stream has finished (hipExtAnyOrderLaunch: no barrier
                                         bit on the dispatch: it starts once the launch in front of it has STARTED); the caller orders producers
                                         and consumers.  monorun_amd.PnPEpnpGroupLaunch uses it for the LM launches of the second and later calls
                                         of a launch set: the first one (ordinary) waits for the set's initialiser, the others run beside it */
#define MR_WAVES_SHIFT        8       /* bits 8..11: wavefronts cooperating on one object (0 = auto, 1,2,4,8) */
#define MR_WAVES_MASK         (0xF << MR_WAVES_SHIFT)
#define MR_LM_MAXIT_SHIFT      16      /* bits 16..21: Ceres' max_num_iterations for the LM (0 = the default, 50; 1..63) */
#define MR_LM_MAXIT_MASK       (0x3F << MR_LM_MAXIT_SHIFT)
#define MR_EPNP_FIRST_ROUND_SHIFT 24   /* mr_epnp_ransac_batched, bits 24..28: hypotheses solved for EVERY object before the replayed RANSAC loop
                                          is consulted (1..30; 0 = the default: 10 for launch sets of fewer than 2048 objects, 3 beyond); the rest are solved only for the objects whose loop still
                                          wants iterations.  Changes the work done, never the result */
#define MR_EPNP_FIRST_ROUND_MASK  (0x1F << MR_EPNP_FIRST_ROUND_SHIFT)
#define MR_EPNP_REFIT_F32   0x40      /* mr_epnp_ransac_batched: normalise the image points of solvePnPRansac's final re-fit in float32 (round 3's
                                         reading of OpenCV) instead of float64 (the published solvePnPRansac converts the inliers to CV_64F first;
                                         the default since round 4) — oracle/epnp.inc "version-dependent decisions" (i) */
#define MR_EPNP_CV_EARLY_RETURN 0x1000 /* mr_epnp_ransac_batched / _grouped: restate OpenCV >= 3.3's early return of solvePnPRansac for EXACTLY FIVE candidates
                                         (`model_points == npoints`: solvePnP(EPNP) on the float32 inputs, no CV_64F conversion -> float32 normalisation
                                         of the image points) instead of the float64 normalisation every re-fit uses (the default; oracle/epnp.inc
                                         decision (ii), orc_set_epnp_cv_early_return).  Four candidates (P3P inside OpenCV) are EPnP either way. */
#define MR_EPNP_DEFER_REFIT 0x80      /* mr_epnp_ransac_batched / _grouped: stop before the last launch (the re-fit's pose candidates on the inliers):
                                         init_mask is final, init_pose / init_valid / diag are NOT written — mr_pnp_uncert_from_epnp_grouped, given
                                         the same (caller-owned, required) workspace on the same stream, does that work as the prologue of the LM
                                         launch, where the object's correspondences are in LDS anyway, and writes them.  Same results, bit for bit. */

/* diag[] layout (per object, 4 floats): */
#define MR_DIAG_LM_ITERATIONS 0       /* LM loop passes executed                                  */
#define MR_DIAG_FINAL_COST    1       /* 1/2 sum r^2 at the returned pose                          */
#define MR_DIAG_WHY           2       /* 1 gradient tol, 2 parameter tol, 3 function tol, 4 max iterations,
                                         5 min radius, 6 invalid steps (failure), 7 evaluation failure,
                                         8 initialiser failed (no LM run)                            */
#define MR_DIAG_WHY_ILL_CONDITIONED 16 /* ADDED to the exit reason when, in some LM pass, the smallest pivot of the Jacobi-scaled damped
                                         normal matrix fell below 1e-10 x the largest: the pose then has a direction the data do not
                                         determine, and a Cholesky step (this library) and Ceres' DENSE_QR step on [J S; D] may return
                                         different iterates (fixture G7b).  reason = value % 16 */
#define MR_DIAG_K0_COUNT      3       /* consensus size of the winning hypothesis (or candidate count) */

int mr_pnp_version(void);
const char *mr_pnp_error_string(int code);
int mr_pnp_last_hip_error(void);
int mr_pnp_device_count(void);
/* waves per object the library picks for `objects_in_flight` objects x P points on the current device (its rule for one launch of that
 * many objects); a caller with several launches in flight passes the objects of ALL of them and puts the answer into MR_WAVES bits.
 * Returns 1, 2 or 4, or a negative MR_ERR_* code. */
int mr_pick_waves(int objects_in_flight, int P);
/* one wavefront busy for `microseconds` on `stream` (stream-overlap self-test of the Python pipeline; asynchronous) */
int mr_spin(int microseconds, void *stream);

/*
 * Batched uncertainty-aware PnP: for each of B objects with P correspondences
 *   candidates  = istd >= istd_thres * mean_P(istd) on both axes  (all points if <= 4 pass)   [R4]
 *   init, mask  = deterministic consensus initialiser K0 (or `init_pose` if given)              [R5]
 *   pose        = trust-region LM (Ceres 1.14 defaults) on the weighted reprojection residuals  [R1,R3]
 *   cov         = inverse(J^T J) at the float32 pose, torch Jacobian masking semantics          [R2,R6,R7]
 *
 * x2d/istd/x3d: device pointers to (B,P,2)/(B,P,2)/(B,P,3) tensors of `in_dtype` addressed as
 *   base[b*strides[0] + p*strides[1] + c*strides[2]] (strides in ELEMENTS).  Both layouts the pipeline
 *   produces are fast paths: channel-planar views of NCHW maps (strides {C*P,1,P}) and contiguous
 *   (B,P,C) (strides {C*P,C,1}).
 * cam_mats: device float (cam_batch,3,3), cam_batch in {1,B}.   u_range/v_range: device float
 *   (range_batch,2), range_batch in {1,B}.   ransac_thr: device float (B) or NULL (no consensus step).
 * init_pose: device double (B,4) [yaw,tx,ty,tz] or NULL.  When given, K0 is skipped and the
 *   candidate set is used as the inlier set.
 * outputs (device): valid (B) u8; pose (B,4) f32 [yaw,tx,ty,tz]; cov (B,16) f32 row-major;
 *   tr_radius (B) f32; inlier_mask (B,P) u8; diag (B,4) f32 or NULL.
 * stream: hipStream_t (NULL = default stream).
 */
int mr_pnp_uncert_batched(
    const void *x2d, const int64_t *x2d_strides,
    const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides,
    int in_dtype,
    const float *cam_mats, int cam_batch,
    const float *u_range, const float *v_range, int range_batch,
    const float *ransac_thr,
    const double *init_pose,
    int B, int P,
    float z_min, float istd_thres, int inlier_opt_only, int flags,
    uint8_t *valid, float *pose, float *cov, float *tr_radius, uint8_t *inlier_mask, float *diag,
    void *stream);

/*
 * The reference's OWN initialiser on the GPU (pnp_uncert_cpu.py:33-68): for each object, on its istd candidates,
 *   cv2.solvePnPRansac(X, x, K, 0, reprojectionError=ransac_thr[b], iterationsCount=max_iters (30), flags=SOLVEPNP_EPNP)
 * — EPnP hypotheses on 5-point subsets drawn by cv::RNG((uint64)-1), float32 squared reprojection error <= thr^2, a model accepted
 * above max(best, 4) inliers, RANSACUpdateNumIters(0.99, ...), EPnP re-fit on the inliers of the best model — or, with ransac_thr
 * NULL, plain cv2.solvePnP(..., SOLVEPNP_EPNP) on the candidates (:54-58).  OpenCV is third-party and not part of the reference
 * tree: the arithmetic is the published algorithm as this repository's test infrastructure restates it.
 * Inputs as for mr_pnp_uncert_batched (x2d / istd / x3d with element strides, cam_mats (cam_batch,3,3), istd_thres, the
 * MR_MEAN_* / MR_NO_ISTD_MASK flags).  Outputs (device): init_pose (B,4) f64 [yaw0 = r_vec[1], tx, ty, tz] (:68; zeros on
 * failure), init_mask (B,P) u8 = the istd candidates narrowed to the RANSAC inliers (:43-51), init_valid (B) u8 (0: RANSAC found no
 * model with 5 inliers, or the pose is not finite — the reference then returns (False, 0, 0, I, 0, mask), :119-125),
 * diag (B,4) f32 or NULL [RANSAC iterations run, inliers of the best model, candidates, index of the best model],
 * debug_hypotheses (B,30,12) f64 or NULL (every hypothesis' R | t; tests).  Feed the three outputs to
 * mr_pnp_uncert_from_init_batched for the LM + covariance.
 * The call is a sequence of six or seven launches on `stream` (sample set-up; speculative hypotheses in two rounds — MR_EPNP_FIRST_ROUND —: a
 * launch that takes a sample to its three candidate poses and one for consensus + OpenCV's sequential loop replayed over the counts; the second
 * round is ONE launch for fewer than 2048 objects — a workgroup per object that leaves at once unless its loop wants more — and the same two
 * compact launches beyond; the re-fit's eigenvectors + beta candidates; the re-fit) that hand their intermediate results over in
 * `workspace`: device memory of at least mr_epnp_workspace_bytes(B, P) bytes, 256-byte aligned, owned by the caller and free to be
 * reused once the work queued on `stream` has passed it (11.5 MB per 1024 objects of 784 correspondences).  workspace = NULL: the library takes it from a
 * stream-ordered memory pool OF ITS OWN (one per device, created on first use, freed blocks kept for the next call: hipMallocFromPoolAsync /
 * hipFreeAsync on `stream`); the process's default pool and its attributes are not touched.  Pass a workspace for steady-state use.
 * Result-neutral tuning knobs, read once per process, for measurements and tests only: MR_EPNP_FIRST_ROUND=n; MR_EP_ROUND2=1 | 2 = the second round always as two
 * launches | always as one; MR_EP_WIDE / MR_EP_WIDE_HYP / MR_EP_WIDE_BETAS = 0 | 2 | 4 = a quad | a 16-lane row | a wave per hypothesis / per re-fit instead of the
 * library's rule by launch size (small launches get rows or waves: the extra quads shorten the matrix's latency chain).  Every setting gives bit-identical outputs
 * (tests/test_gpu_epnp.py::test_wide_launches_equal_the_quad_launches).
 * VERSION-DEPENDENT DECISIONS (oracle/epnp.inc): with thresholds, an object with exactly FOUR (only possible when P = 4) or FIVE candidates
 * deliberately differs from OpenCV >= 3.3 as published, whose solvePnPRansac returns early when model_points == npoints (P3P's pose for
 * four points, float32 EPnP for five): here both get EPnP on the candidates with the float64 normalisation of every re-fit (MR_EPNP_CV_EARLY_RETURN restates the five-candidate early return), all of them
 * inliers; P3P and its solvability test are not restated.  P >= 6 candidates (every shipped configuration) is unaffected.
 * MR_EPNP_REFIT_F32 selects round 3's float32 re-fit.
 */
int mr_epnp_ransac_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *ransac_thr, int B, int P,
    float istd_thres, int flags, int max_iters,
    double *init_pose, uint8_t *init_mask, uint8_t *init_valid, float *diag, double *debug_hypotheses,
    void *workspace, size_t workspace_bytes, void *stream);
size_t mr_epnp_workspace_bytes(int B, int P);

/*
 * The same initialiser for SEVERAL calls in one launch set (1 <= ncalls <= 8): call c has its own correspondence tensors x2d[c] /
 * istd[c] / x3d[c], camera cam_mats[c], thresholds ransac_thr[c] and outputs init_pose[c] / init_mask[c] / init_valid[c] / diag[c]
 * (arrays of ncalls device pointers, read on the host); B objects per call, and P, the element strides, in_dtype, cam_batch, istd_thres,
 * flags and max_iters are common (ransac_thr and diag: all NULL or none).  Results are those of ncalls calls of mr_epnp_ransac_batched,
 * bit for bit.  Why it exists: HIP runs the launches of at most four streams side by side, and every stage of this initialiser is a
 * latency chain that fills a fraction of the chip — a launch that carries the objects of two calls keeps eight calls' stages in flight on
 * four streams (monorun_amd.PnPEpnpGroupLaunch; measured on MI355X: DESIGN.md section 3).  workspace: at least
 * mr_epnp_workspace_bytes(ncalls * B, P) bytes (NULL: the library's pool, as above).
 */
int mr_epnp_ransac_grouped(
    int ncalls, const void *const *x2d, const int64_t *x2d_strides, const void *const *istd, const int64_t *istd_strides,
    const void *const *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *const *cam_mats, int cam_batch, const float *const *ransac_thr, int B, int P,
    float istd_thres, int flags, int max_iters,
    double *const *init_pose, uint8_t *const *init_mask, uint8_t *const *init_valid, float *const *diag,
    void *workspace, size_t workspace_bytes, void *stream);

/*
 * LM + covariance (stages 3 and 4 of mr_pnp_uncert_batched) from an EXTERNAL initialiser's result: init_mask (B,P) u8 is the
 * candidate = inlier set the LM sees (inlier_opt_only) and the mask the covariance uses, init_pose (B,4) f64 the start, init_valid
 * (B) u8 = 0 marks objects whose initialiser failed (valid = 0, zero pose, as pnp_uncert_cpu.py:119-125).  Everything else as
 * for mr_pnp_uncert_batched; inlier_mask (B,P) receives init_mask.
 */
int mr_pnp_uncert_from_init_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    const double *init_pose, const uint8_t *init_mask, const uint8_t *init_valid, int B, int P,
    float z_min, int inlier_opt_only, int flags,
    uint8_t *valid, float *pose, float *cov, float *tr_radius, uint8_t *inlier_mask, float *diag, void *stream);

/*
 * The same launch over the objects of SEVERAL calls (1 <= ncalls <= 8; the companion of mr_epnp_ransac_grouped): arrays of ncalls device
 * pointers (read on the host) for everything a caller owns; B objects per call; P, strides, in_dtype, cam_batch, range_batch, z_min and
 * the flags are common (inlier_mask and diag: all NULL or none).  Results are those of ncalls calls of mr_pnp_uncert_from_init_batched,
 * bit for bit.  One launch lasts as long as its slowest object: carried by one launch, the calls of a launch set pay that tail once.
 */
int mr_pnp_uncert_from_init_grouped(
    int ncalls, const void *const *x2d, const int64_t *x2d_strides, const void *const *istd, const int64_t *istd_strides,
    const void *const *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *const *cam_mats, int cam_batch, const float *const *u_range, const float *const *v_range, int range_batch,
    const double *const *init_pose, const uint8_t *const *init_mask, const uint8_t *const *init_valid, int B, int P,
    float z_min, int inlier_opt_only, int flags,
    uint8_t *const *valid, float *const *pose, float *const *cov, float *const *tr_radius, uint8_t *const *inlier_mask, float *const *diag, void *stream);

/*
 * mr_pnp_uncert_from_init_grouped for an initialiser that ran with MR_EPNP_DEFER_REFIT (1 <= ncalls <= 8; one call is a launch set of
 * one): the LM launch loads each object's correspondences ONCE and runs the initialiser's last step — the three pose candidates of the
 * re-fit on the inliers, pnp_uncert_cpu.py:52-57's cv2.solvePnP inside solvePnPRansac — before the LM.  init_pose (B,4) f64, init_valid
 * (B) u8 and epnp_diag (B,4) f32 (array or entries NULL: none) are OUTPUTS here, with the values mr_epnp_ransac_* would have written;
 * init_mask is the initialiser's.  workspace = the one given to mr_epnp_ransac_* (same B, P, ncalls), untouched in between.  One launch,
 * one pass over the correspondences and one workgroup residency less per call than the two entry points one after the other; results
 * identical, bit for bit.  cov_calib (array of ncalls (B,16) f32 outputs, or NULL / all entries NULL: none) receives the calibrated and
 * distance-corrected covariance of mr_pnp_from_head_batched's epilogue: (s s^T) * cov with s = exp(cov_calib_logscale[0..3]) (device
 * pointer, read at run time; uncert_prop_pnp_optimizer.py:96-97), times (cov_corr_sd / ||t||)^2 when cov_corr_sd > 0
 * (monorun_roi_head.py:530-534) — float32, the torch operation order.
 */
int mr_pnp_uncert_from_epnp_grouped(
    int ncalls, const void *const *x2d, const int64_t *x2d_strides, const void *const *istd, const int64_t *istd_strides,
    const void *const *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *const *cam_mats, int cam_batch, const float *const *u_range, const float *const *v_range, int range_batch,
    double *const *init_pose, const uint8_t *const *init_mask, uint8_t *const *init_valid, float *const *epnp_diag, int B, int P,
    float z_min, int inlier_opt_only, int flags,
    uint8_t *const *valid, float *const *pose, float *const *cov, float *const *tr_radius, uint8_t *const *inlier_mask, float *const *diag,
    const float *cov_calib_logscale, float cov_corr_sd, float *const *cov_calib,
    const void *workspace, size_t workspace_bytes, void *stream);

/*
 * The reference's eigenvalue rule for ill-conditioned Hessians (pnp_uncert.py:77-85), applied per object to the outputs of
 * mr_pnp_uncert_batched / mr_pnp_uncert_from_init_batched: an object stays valid only if lambda_min(h) > max(1e-6 * lambda_max(h), 0)
 * (evaluated on cov = h^-1, whose eigenvalues are the reciprocals); otherwise valid[b] = 0 and cov[b] = identity.  The fused kernel
 * alone invalidates only objects whose h has no Cholesky factorisation; the reference applies this rule — to the whole batch — in
 * the branch it takes when torch.inverse raises.  valid (B) u8 in/out, cov (B,16) f32 in/out, eig_min_max (B,2) f32 or NULL (the
 * two extreme eigenvalues of cov, for inspection).
 */
int mr_cov_symeig_rule(uint8_t *valid, float *cov, int B, float *eig_min_max, void *stream);

/*
 * True 6-DoF refinement (the flag the reference declares and ignores: `use_6dof`, pnp_uncert.py:11).
 * Second launch of pnp_uncert(..., use_6dof=True): for each object, starting from the 4-DoF result pose4 = [yaw,tx,ty,tz]
 * (r = (0,yaw,0)) on the points of that solve's final inlier_mask, the same residual functor (pnp_uncert_cpu.cpp:24-51) is
 * minimised over pose6 = [rx,ry,rz,tx,ty,tz] (angle-axis) with the same Ceres-1.14 LM; cov6 = (J^T J)^-1 (6x6, row-major)
 * with the solver's Jacobian at the returned pose.  Inputs / strides / cameras / ranges as for mr_pnp_uncert_batched;
 * inlier_mask (B,P) u8, pose4 (B,4) f32, valid4 (B) u8 = outputs of that call.  Outputs: valid (B) u8, pose6 (B,6) f32,
 * cov6 (B,36) f32 (identity when invalid), diag (B,2) f32 [LM iterations, exit reason as MR_DIAG_WHY] or NULL.
 * flags: only the MR_LM_MAXIT bits are read.
 */
int mr_pnp6_refine_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    const uint8_t *inlier_mask, const float *pose4, const uint8_t *valid4, int B, int P, float z_min, int flags,
    uint8_t *valid, float *pose6, float *cov6, float *diag, void *stream);

/*
 * `forward_exact_hessian=True` (the reference's hessian.py:5-64 used by pnp_uncert.py:63-85): second launch after
 * mr_pnp_uncert_batched (which may then run with MR_COV_NONE).  h[i][j] = d (J^T e)_i / d pose_j of the masked, weighted
 * reprojection cost at `pose` (B,4) f32 [yaw,tx,ty,tz] — masks (z clip, per-axis uv clip, points outside inlier_mask) are
 * constants, as they are for the reference's autograd — and cov = inverse(h) (LU semantics: h need not be positive definite).
 * Inputs / strides / cameras / ranges as for mr_pnp_uncert_batched; inlier_mask (B,P) u8 or NULL (all points).
 * valid (B) u8 is IN/OUT: objects entering with 0 get h = 0, cov = identity; an exactly singular or non-finite h clears the
 * flag and yields cov = identity (the per-object reading of pnp_uncert.py:79-85).  hess (B,16) f32 may be NULL; cov (B,16) f32.
 */
int mr_pnp_exact_hessian_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    const float *pose, const uint8_t *inlier_mask, int B, int P, float z_min,
    uint8_t *valid, float *hess, float *cov, void *stream);

/*
 * The reference's own per-object C entry point, same signature and semantics (ext.h:1-13,
 * pnp_uncert_cpu.cpp:245-292): HOST fp64 buffers in, host results out; runs the same LM kernel on the
 * GPU for one object (blocking).  result_cov may be NULL; on failure result_cov is left untouched.
 */
void pnp_uncert(double *pts2d, double *pts3d, double *wgt2d, double *K, double *init_pose,
                int *result_val, double *result_pose, double *result_cov, double *result_tr,
                int pn, double *clips);

/*
 * Fused NOC-head post-processing ("K2"): from the raw head output to the PnP-boundary tensors.
 *   all_pred  (B, 2*C*5, h, w) f32 / f16 / bf16 (pred_dtype; converted exactly to f32, all arithmetic is f32) — conv_final
 *             output (C = num_classes, or 1 if class_agnostic)
 *   labels (B) int64, flip (B) u8, dim (B,3) f32 normalised dims, dim_var (B,3) f32 or NULL,
 *   rois (B,4) f32 xyxy in the test-scale image
 * writes channel-planar maps
 *   coords_2d (B,2,h,w), coords_2d_istd (B,2,h,w), coords_3d (B,3,h,w)   [all f32]
 *   dims (B,3), dims_var (B,3) (optional), ransac_thr (B) (optional)
 * coords_2d = roi_align(coord_2d, rois, (h,w), 1.0, 0, 'avg', True) (monorun_roi_head.py:521-523).  With
 *   coord_2d_map == NULL the map is taken to be the identity pixel grid of the image the RoIs live in (test configs:
 *   scale_factor 1.0, no flip) and the bin centres are written analytically; with a (2,H,W) map (loading.py:67-78 after
 *   Resize3D / RandomFlip3D / Pad3D) the RoIAlign taps are sampled exactly, mmcv border rules included.
 */
int mr_noc_decode_batched(
    const void *all_pred, int pred_dtype /* MR_F32 | MR_F16 | MR_BF16 */, const int64_t *labels, const uint8_t *flip,
    const float *dim, const float *dim_var, const float *rois,
    int B, int num_classes, int class_agnostic, int h, int w,
    const float *dim_means /* (C,3) */, const float *dim_stds /* (C,3) */,
    const float *noc_means /* 3 */, const float *noc_stds /* 3 */,
    double proj_scaling_denominator /* ref_length*ref_focal_y*target_std, e.g. 173.28 */, double ref_focal_y, double epistemic_std_gain,
    float std_scale, float ransac_thres_ratio /* <0: ransac_thr not written */,
    float *coords_2d, float *coords_2d_istd, float *coords_3d,
    float *dims, float *dims_var, float *ransac_thr,
    const float *coord_2d_map /* (2,H,W) device map or NULL */, int map_h, int map_w,
    void *stream);

/*
 * Fused K2 + PnP: from the raw NOC-head output to the pose in ONE launch — the decoded maps are built
 * directly in the kernel's LDS tile and never touch HBM.  Arguments = those of mr_noc_decode_batched (head
 * output, constants) followed by those of mr_pnp_uncert_batched (camera, ranges, options, outputs); results
 * are identical, bit for bit, to calling the two entry points in sequence.  dims / dims_var (B,3): optional
 * decoded dimensions for the consumers.  ransac_thres_ratio < 0 disables the consensus step.
 * cov_calib (optional): the pose head's calibrated covariance (s s^T) * cov with s = exp(cov_calib_logscale)
 * (uncert_prop_pnp_optimizer.py:96-97), times (cov_corr_sd / ||t||)^2 when cov_corr_sd > 0 (cov_correction,
 * distance_invar_proj_error_coder.py:62-63, monorun_roi_head.py:530-534) — written by the same launch.
 */
int mr_pnp_from_head_batched(
    const void *all_pred, int pred_dtype, const int64_t *labels, const uint8_t *flip,
    const float *dim, const float *dim_var, const float *rois,
    int B, int num_classes, int class_agnostic, int h, int w,
    const float *dim_means, const float *dim_stds, const float *noc_means, const float *noc_stds,
    double proj_scaling_denominator, double ref_focal_y, double epistemic_std_gain,
    float std_scale, float ransac_thres_ratio,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    float z_min, float istd_thres, int inlier_opt_only, int flags,
    uint8_t *valid, float *pose, float *cov, float *tr_radius, uint8_t *inlier_mask, float *diag,
    float *dims, float *dims_var, const float *coord_2d_map, int map_h, int map_w,
    const float *cov_calib_logscale /* (4) or NULL */, float cov_corr_sd /* <= 0: no distance correction */,
    float *cov_calib /* (B,16) or NULL */, void *stream);

/*
 * N3: RoIAlign forward, average pooling — mmcv.ops.roi_align(input, rois, (out_h,out_w), spatial_scale, sampling_ratio,
 * 'avg', aligned) as called at monorun_roi_head.py:521-523 (mmcv is third-party and absent; published algorithm).
 * input (N,C,H,W) f32, rois (K,5) [batch_idx, x1, y1, x2, y2], output (K,C,out_h,out_w).  sampling_ratio 0 = adaptive.
 */
int mr_roi_align_avg(const float *input, const float *rois, int K, int C, int H, int W, int out_h, int out_w,
                     float spatial_scale, int sampling_ratio, int aligned, float *output, void *stream);

/*
 * N1: rotated-BEV NMS of the pose consumers — replaces mmdet3d.ops.iou3d.nms_gpu as called by
 * multiclass_3d_result_nms (monorun/models/roi_heads/monorun_roi_head.py:619-655).
 *   boxes_xyxyr (total,5) f32 [x1, y1, x2, y2, ry] (xywhr2xyxyr, :657-677), scores (total) f32,
 *   offsets (groups+1) i32: group g owns rows [offsets[g], offsets[g+1]) (one group per class), max_group =
 *   largest group size (<= 512).  A box is suppressed when its rotated IoU with an already kept, higher
 *   scoring box of its group exceeds thr (score ties: lower index first).
 *   keep (total) i64: kept indices LOCAL to the group, descending score, written at keep[offsets[g] ...];
 *   num_keep (groups) i32.
 */
int mr_nms_bev_batched(const float *boxes_xyxyr, const float *scores, const int32_t *offsets, int groups, int max_group,
                       float thr, int64_t *keep, int32_t *num_keep, void *stream);

/*
 * N2: KITTI object evaluator, device side — replaces the numba / numba-CUDA code of
 * monorun/core/evaluation/kitti_utils/eval.py and rotate_iou.py.  All arrays are device pointers; images are addressed
 * through exclusive prefix offsets (n_img+1 entries).  Box rows are double[12]:
 *   x1 y1 x2 y2 (2-D box) | x y z (location, y = bottom) | l h w (dimensions) | rotation_y | score
 *
 * mr_kitti_overlaps: the per-image overlap blocks that eval_class reads (eval.py:477-479 -> calculate_iou_partly,
 *   eval.py:341-416): overlaps[ov_off[i] + j*n_gt_i + k] for detection j and label k of image i.
 *   metric 0 = image_box_overlap (eval.py:84-112), 1 = bev_box_overlap (rotate_iou.py:256-281, inputs rounded to float32,
 *   result rounded to float32), 2 = d3_box_overlap (eval.py:121-158).  arith32: the annotation arrays were float32, so
 *   numba's arithmetic was float32;  out32: results are stored in a float32 array.
 */
int mr_kitti_overlaps(int metric, int arith32, int out32, int n_img, const int64_t *dt_off, const int64_t *gt_off,
                      const int64_t *ov_off, int64_t total_pairs, const double *dt_box, const double *gt_box,
                      double *overlaps, void *stream);

/*
 * mr_kitti_match: compute_statistics_jit (eval.py:161-279) for every image and every "combo" = (class, difficulty,
 *   min_overlap) at once.  ign_gt / ign_dt: int8 [n_cd][total] rows of clean_data's ignore codes (eval.py:28-80), one row
 *   per (class, difficulty); combo_cd[c] selects the row, combo_min_overlap[c] the IoU threshold.
 *   second_pass = 0 (compute_fp=False, eval.py:499-513): match_score [n_combo][total_gt] receives the score of the
 *     detection matched to each true-positive label (entries of other labels are left untouched: pre-fill with NaN).
 *   second_pass = 1 (fused_compute_statistics, eval.py:291-338): thresholds [n_combo][41] with n_thr[c] valid entries;
 *     pr [n_combo][41][4] receives (tp, fp, fn, similarity) summed over the images in image order.  dc_box (total_dc,4) /
 *     dc_off: DontCare regions (metric 0).  workspace: >= mr_kitti_match_workspace_bytes(n_img, n_combo) bytes.
 *   alpha32 / dtdata32: the label+detection / detection arrays were float32 (numba's alpha difference and the DontCare
 *   overlap are then float32).  max_det: largest number of detections in one image (<= 512).
 */
int64_t mr_kitti_match_workspace_bytes(int n_img, int n_combo);
int mr_kitti_match(int second_pass, int metric, int compute_aos, int alpha32, int dtdata32, int n_img, int max_det,
                   const int64_t *dt_off, const int64_t *gt_off, const int64_t *ov_off, const int64_t *dc_off,
                   int64_t total_dt, int64_t total_gt,
                   const double *overlaps, const double *dt_box, const double *dt_alpha, const double *gt_alpha,
                   const double *dc_box, const int8_t *ign_gt, const int8_t *ign_dt,
                   int n_combo, const int32_t *combo_cd, const double *combo_min_overlap,
                   const double *thresholds, const int32_t *n_thr, double *match_score, double *pr,
                   void *workspace, int64_t workspace_bytes, void *stream);

/*
 * N4: the two further entry points the reference's ext.h declares (ext.h:15-43,
 * pnp_uncert_cpu.cpp:294-377) — exported by the reference, never called by its Python code.  Same signatures
 * and semantics: dimpose = [log l, log h, log w, yaw, tx, ty, tz]; pts3d are NOC coordinates scaled by
 * exp(log-dims); every residual block goes through one Huber loss (delta); wgt2d is (n,2) [wxx, wyy] for
 * pnp_noc_uncert and (n,3) [wxx, wxy, wyy] for pnp_noc_cov_uncert.  Host fp64 buffers, one object, blocking.
 */
void pnp_noc_uncert(double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt, double *K,
                    double *init_dimpose, int *result_val, double *result_dimpose, int pn, double *clips, double delta);
void pnp_noc_cov_uncert(double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt, double *K,
                        double *init_dimpose, int *result_val, double *result_dimpose, int pn, double *clips, double delta);

/*
 * Batched form of the two solvers above: B objects with pn correspondences each, DEVICE fp64 buffers, one workgroup per
 * object, asynchronous on `stream`.  full_cov 0 = pnp_noc_uncert (wgt2d (B,pn,2)), 1 = pnp_noc_cov_uncert (wgt2d (B,pn,3)).
 * pts2d (B,pn,2), pts3d (B,pn,3) NOC coordinates, logdim / logdim_wgt (B,3), K (K_batch,9) and clips (clips_batch,5) with
 * batch 1 (shared) or B, init_dimpose (B,7).  Outputs: result_dimpose (B,7), result_val (B) int32 in {0,1}, diag (B,2)
 * [LM iterations, final robustified cost] or NULL.
 */
int mr_pnp_noc_batched(int full_cov, const double *pts2d, const double *pts3d, const double *wgt2d, const double *logdim,
                       const double *logdim_wgt, const double *K, int K_batch, const double *init_dimpose, const double *clips,
                       int clips_batch, double delta, int B, int pn, double *result_dimpose, int32_t *result_val, double *diag,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MONORUN_PNP_H_ */

/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
 *
 * Plain-C, scalar, fp64 CPU restatement of MonoRUn's uncertainty-aware 4-DoF PnP hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * monorun_amd/ never imports, links or calls anything in oracle/.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   R1  residual + autodiff Jacobian ........ monorun/ops/least_squares/src/pnp_uncert_cpu.cpp:24-51
 *        (Jet arithmetic + AngleAxisRotatePoint: Ceres Solver 1.14.0 jet.h / rotation.h —
 *         third-party, NOT in the reference tree, pinned by INSTALL.md:29; restated from the
 *         published algorithm)
 *   R3  LM driver, options, outputs .......... pnp_uncert_cpu.cpp:245-292  (+ Ceres 1.14
 *        TrustRegionMinimizer / LevenbergMarquardtStrategy defaults, restated)
 *   R2/R7 torch-semantics Jacobian and J^T J . monorun/ops/least_squares/jacobian.py:4-98,141-167,
 *        hessian.py:67-87
 *   R6  pose covariance inverse(J^T J) ....... monorun/ops/least_squares/pnp_uncert.py:60-85
 *   R5  per-object driver .................... monorun/ops/least_squares/pnp_uncert_cpu.py:11-125
 *   K0  initialiser: the reference calls cv2.solvePnPRansac/solvePnP(EPNP) (pnp_uncert_cpu.py:35-58);
 *        OpenCV is absent everywhere, so K0 here restates THIS repo's documented deterministic
 *        replacement, not OpenCV.
 *
 * PARITY STATUS: residual/Jacobian/J^T J/covariance are pinned against golden vectors produced by
 * importing the reference's jacobian.py / hessian.py (tests/golden/make_golden.py).  The LM
 * *stopping iterate* follows Ceres 1.14 from its documentation/recollection of its source and is
 * "parity unpinned" against a real Ceres binary (none can be built here); the LM *solution* is
 * pinned against scipy.optimize.least_squares and noise-free known-answer cubes.  K0 is unpinned
 * against OpenCV by construction.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off, the reference's own -O2; no -march).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <float.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * Forward-mode dual numbers (Ceres Jet arithmetic): oracle/jet.inc instantiated for 4 partials
 * (pose) and 7 partials (log-dims + pose, the N4 variants).
 * ---------------------------------------------------------------------------------------- */
#define JN 4
#define JET jet4
#define JF(name) j_##name
#include "jet.inc"
#undef JN
#undef JET
#undef JF
#define JN 7
#define JET jet7
#define JF(name) j7_##name
#include "jet.inc"
#undef JN
#undef JET
#undef JF
#define JN 6
#define JET jet6
#define JF(name) j6_##name
#include "jet.inc"
#undef JN
#undef JET
#undef JF

/* ------------------------------------------------------------------------------------------
 * R1: ReprojectionErrorArray::operator() on Jets (pnp_uncert_cpu.cpp:24-51).
 * pose = [yaw, tx, ty, tz].  res[2], jac[2][4] (row-major).  Clamped quantities are replaced by
 * constants, i.e. lose all their partials (z-clamp: only z; u/v clamp: the whole projected coord).
 * ---------------------------------------------------------------------------------------- */
typedef struct { double fx, fy, cx, cy, z_min, u_min, u_max, v_min, v_max; } orc_cam;

static void orc_residual_jet(const orc_cam *c, const double pose[4],
                             double x2d, double y2d, double x3d, double y3d, double z3d,
                             double wxx, double wyy, double res[2], double jac[8]) {
    jet4 p[4] = { j_var(pose[0], 0), j_var(pose[1], 1), j_var(pose[2], 2), j_var(pose[3], 3) };
    jet4 pts3d[3] = { j_const(x3d), j_const(y3d), j_const(z3d) };
    jet4 r_vec[3] = { j_const(0.0), p[0], j_const(0.0) };
    jet4 t[3];
    j_angle_axis_rotate_point(r_vec, pts3d, t);
    t[0] = j_add(t[0], p[1]);
    t[1] = j_add(t[1], p[2]);
    t[2] = j_add(t[2], p[3]);
    /* std::max(a, b) == (a < b) ? b : a on the scalar parts  (pnp_uncert_cpu.cpp:36) */
    if (t[2].a < c->z_min) t[2] = j_const(c->z_min);
    jet4 proj_x = j_add(j_div(j_mul(j_const(c->fx), t[0]), t[2]), j_const(c->cx));
    jet4 proj_y = j_add(j_div(j_mul(j_const(c->fy), t[1]), t[2]), j_const(c->cy));
    if (proj_x.a < c->u_min) proj_x = j_const(c->u_min); else if (proj_x.a > c->u_max) proj_x = j_const(c->u_max);
    if (proj_y.a < c->v_min) proj_y = j_const(c->v_min); else if (proj_y.a > c->v_max) proj_y = j_const(c->v_max);
    jet4 dx = j_sub(proj_x, j_const(x2d));
    jet4 dy = j_sub(proj_y, j_const(y2d));
    jet4 r0 = j_mul(j_const(wxx), dx);
    jet4 r1 = j_mul(j_const(wyy), dy);
    res[0] = r0.a; res[1] = r1.a;
    for (int i = 0; i < 4; ++i) { jac[i] = r0.v[i]; jac[4 + i] = r1.v[i]; }
}

/* exported for the golden-vector tests */
void orc_residual_jacobian(const double *K, const double *clips, const double *pose,
                           const double *pts2d, const double *pts3d, const double *wgt2d, int pn,
                           double *res /*pn,2*/, double *jac /*pn,2,4*/) {
    orc_cam c = { K[0], K[4], K[2], K[5], clips[0], clips[1], clips[2], clips[3], clips[4] };
    for (int i = 0; i < pn; ++i)
        orc_residual_jet(&c, pose, pts2d[2 * i], pts2d[2 * i + 1], pts3d[3 * i], pts3d[3 * i + 1],
                         pts3d[3 * i + 2], wgt2d[2 * i], wgt2d[2 * i + 1], res + 2 * i, jac + 8 * i);
}

/* ------------------------------------------------------------------------------------------
 * Problem evaluation: cost = 1/2 sum r^2, g = J^T r, H = J^T J  (what Ceres' evaluator hands the
 * minimizer).  Returns 0 when anything is non-finite (Ceres: "evaluation failed").
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    orc_cam cam; int pn;
    const double *pts2d, *pts3d, *wgt2d;
} orc_problem;

static int orc_eval(const orc_problem *pb, const double x[4], double *cost, double g[4], double H[16]) {
    double c = 0.0; double gg[4] = {0, 0, 0, 0}; double HH[16]; memset(HH, 0, sizeof HH);
    for (int i = 0; i < pb->pn; ++i) {
        double r[2], J[8];
        orc_residual_jet(&pb->cam, x, pb->pts2d[2 * i], pb->pts2d[2 * i + 1], pb->pts3d[3 * i],
                         pb->pts3d[3 * i + 1], pb->pts3d[3 * i + 2], pb->wgt2d[2 * i], pb->wgt2d[2 * i + 1], r, J);
        c += r[0] * r[0] + r[1] * r[1];
        if (g) for (int a = 0; a < 4; ++a) {
            gg[a] += J[a] * r[0] + J[4 + a] * r[1];
            for (int b = 0; b < 4; ++b) HH[4 * a + b] += J[a] * J[b] + J[4 + a] * J[4 + b];
        }
    }
    *cost = 0.5 * c;
    int ok = isfinite(*cost);
    if (g) { for (int a = 0; a < 4; ++a) { g[a] = gg[a]; ok = ok && isfinite(gg[a]); }
             for (int a = 0; a < 16; ++a) { H[a] = HH[a]; ok = ok && isfinite(HH[a]); } }
    return ok;
}

/* Cholesky solve of a symmetric n x n (n<=7) system, row-major full storage. 0 on non-PD.
 * Reciprocal form: one sqrt and one division per pivot, multiplications elsewhere, explicit fma —
 * the exact operation sequence is part of the K0 specification. */
static int orc_chol_solve(int n, const double *A, const double *b, double *x) {
    double L[49], inv[7];
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s = fma(-L[j * n + k], L[j * n + k], s);
        if (!(s > 0.0) || !isfinite(s)) return 0;
        const double d = sqrt(s);
        L[j * n + j] = d; inv[j] = 1.0 / d;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t = fma(-L[i * n + k], L[j * n + k], t);
            L[i * n + j] = t * inv[j];
        }
    }
    double y[7];
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s = fma(-L[i * n + k], y[k], s); y[i] = s * inv[i]; }
    for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s = fma(-L[k * n + i], x[k], s); x[i] = s * inv[i]; }
    return 1;
}

/* inverse of SPD 4x4 via Cholesky (column by column). 0 on failure. */
static int orc_spd_inverse4(const double H[16], double inv[16]) {
    for (int c = 0; c < 4; ++c) {
        double e[4] = {0, 0, 0, 0}, x[4]; e[c] = 1.0;
        if (!orc_chol_solve(4, H, e, x)) return 0;
        for (int r = 0; r < 4; ++r) inv[4 * r + c] = x[r];
    }
    for (int i = 0; i < 16; ++i) if (!isfinite(inv[i])) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * R3: Ceres 1.14 trust-region Levenberg-Marquardt, restated (defaults; only
 * linear_solver_type=DENSE_QR is set by the reference, pnp_uncert_cpu.cpp:270-271).
 * The dense-QR solve of  min ||[J;D] y - [r;0]||  is done through its normal equations
 * (J^T J + D^2) y = J^T r  (mathematically identical; 4x4, Jacobi-scaled, fp64).
 * ---------------------------------------------------------------------------------------- */
enum { ORC_CONVERGENCE = 0, ORC_NO_CONVERGENCE = 1, ORC_FAILURE = 2 };
enum { ORC_WHY_GRADIENT = 1, ORC_WHY_PARAMETER = 2, ORC_WHY_FUNCTION = 3, ORC_WHY_MAXITER = 4,
       ORC_WHY_MINRADIUS = 5, ORC_WHY_INVALID = 6, ORC_WHY_EVALFAIL = 7 };

typedef struct {
    int termination, why, num_iterations /* loop passes executed (step attempts) */;
    int num_successful;
    double initial_cost, final_cost, radius;
} orc_lm_summary;

typedef int (*orc_eval_fn)(const void *ctx, const double *x, double *cost, double *g, double *H);
/* residuals r[nres] and Jacobian J[nres][n] (row-major) at x — needed only by the DENSE_QR step solver */
typedef void (*orc_jac_fn)(const void *ctx, const double *x, double *r, double *J);
#define ORC_MAXN 7

/* Options of one LM run.  qr = 0: the trust-region step through the Jacobi-scaled normal equations (Cholesky) — what the
 * HIP kernel does;  qr = 1: literally what Ceres' DENSE_QR does (dense_qr_solver.cc: Householder QR of the augmented
 * (m+n) x n matrix [J S; D], right-hand side [r; 0]) and the model cost change from the model residuals J S step
 * (trust_region_minimizer.cc).  max_iter: Ceres' max_num_iterations (default 50).  trace: optional per-pass record. */
#define ORC_TRACE_W 8   /* iteration, cost, candidate cost, model cost change, relative decrease, radius after, step norm, outcome */
typedef struct { int qr, max_iter; double *trace; int trace_cap, trace_n; } orc_lm_opts;
static orc_lm_opts orc_default_opts = { 0, 50, NULL, 0, 0 };
void orc_set_lm_options(int qr, int max_iter) { orc_default_opts.qr = qr; orc_default_opts.max_iter = max_iter > 0 ? max_iter : 50; }

/* least squares min ||A y - b|| by Householder reflections (no pivoting), A is m x n row-major (destroyed), b[m] (destroyed) */
static int orc_householder_ls(int m, int n, double *A, double *b, double *y) {
    for (int k = 0; k < n; ++k) {
        double nrm = 0.0;
        for (int i = k; i < m; ++i) nrm += A[n * i + k] * A[n * i + k];
        nrm = sqrt(nrm);
        if (!(nrm > 0.0) || !isfinite(nrm)) return 0;
        const double alpha = A[n * k + k] > 0.0 ? -nrm : nrm;
        const double v0 = A[n * k + k] - alpha;
        /* v = (v0, A[k+1.., k]); H = I - 2 v v^T / (v^T v) */
        double vtv = v0 * v0;
        for (int i = k + 1; i < m; ++i) vtv += A[n * i + k] * A[n * i + k];
        if (vtv > 0.0) {
            for (int j = k + 1; j < n; ++j) {
                double d = v0 * A[n * k + j];
                for (int i = k + 1; i < m; ++i) d += A[n * i + k] * A[n * i + j];
                const double f = 2.0 * d / vtv;
                A[n * k + j] -= f * v0;
                for (int i = k + 1; i < m; ++i) A[n * i + j] -= f * A[n * i + k];
            }
            double d = v0 * b[k];
            for (int i = k + 1; i < m; ++i) d += A[n * i + k] * b[i];
            const double f = 2.0 * d / vtv;
            b[k] -= f * v0;
            for (int i = k + 1; i < m; ++i) b[i] -= f * A[n * i + k];
        }
        A[n * k + k] = alpha;
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < n; ++j) s -= A[n * i + j] * y[j];
        y[i] = s / A[n * i + i];
    }
    return 1;
}

static int orc_lm_iter0_gradient_test = 1;
void orc_set_lm_iter0_gradient_test(int on) { orc_lm_iter0_gradient_test = on ? 1 : 0; }
static void orc_lm_n_ex(int n, orc_eval_fn ev, orc_jac_fn jf, int nres, const void *ctx, const double *init, double *out,
                        orc_lm_summary *sm, orc_lm_opts *op) {
    const int    max_num_iterations = op->max_iter;
    const double initial_radius = 1e4, max_radius = 1e16, min_radius = 1e-32;
    const double min_relative_decrease = 1e-3;
    const double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const int    max_consecutive_invalid = 5;

    double x[ORC_MAXN]; memcpy(x, init, sizeof(double) * n); memcpy(out, init, sizeof(double) * n);
    double cost, g[ORC_MAXN], H[ORC_MAXN * ORC_MAXN];
    double radius = initial_radius, decrease_factor = 2.0;
    memset(sm, 0, sizeof *sm); sm->radius = radius;

    /* IterationZero */
    if (!ev(ctx, x, &cost, g, H)) { sm->termination = ORC_FAILURE; sm->why = ORC_WHY_EVALFAIL; sm->radius = 0.0; return; }
    sm->initial_cost = sm->final_cost = cost;
    double scale[ORC_MAXN];
    for (int j = 0; j < n; ++j) scale[j] = 1.0 / (1.0 + sqrt(H[(n + 1) * j]));   /* jacobi_scaling, from the initial J */
    double x_norm = 0.0; for (int j = 0; j < n; ++j) x_norm += x[j] * x[j]; x_norm = sqrt(x_norm);
    int last_successful = orc_lm_iter0_gradient_test;   /* iteration 0 counts as successful (IterationZero sets it): version-dependent decision (iii), epnp.inc header */
    int iteration = 0, invalid_run = 0;

    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue (parameters are committed here) */
        if (last_successful) { memcpy(out, x, sizeof(double) * n); sm->final_cost = cost; }
        sm->radius = radius; sm->num_iterations = iteration;
        if (iteration >= max_num_iterations) { sm->termination = ORC_NO_CONVERGENCE; sm->why = ORC_WHY_MAXITER; return; }
        if (last_successful) {
            double gmax = 0.0; for (int j = 0; j < n; ++j) if (fabs(g[j]) > gmax) gmax = fabs(g[j]);
            if (gmax <= gradient_tolerance) { sm->termination = ORC_CONVERGENCE; sm->why = ORC_WHY_GRADIENT; return; }
        }
        if (radius <= min_radius) { sm->termination = ORC_CONVERGENCE; sm->why = ORC_WHY_MINRADIUS; return; }

        ++iteration; last_successful = 0;
        const double trace_cost = cost;
        /* ComputeTrustRegionStep — LevenbergMarquardtStrategy::ComputeStep on the scaled Jacobian */
        double Hs[ORC_MAXN * ORC_MAXN], gs[ORC_MAXN], A[ORC_MAXN * ORC_MAXN], D2[ORC_MAXN], y[ORC_MAXN], step[ORC_MAXN];
        for (int a = 0; a < n; ++a) { gs[a] = g[a] * scale[a]; for (int b = 0; b < n; ++b) Hs[n * a + b] = H[n * a + b] * scale[a] * scale[b]; }
        for (int j = 0; j < n; ++j) {
            double d = Hs[(n + 1) * j]; d = fmin(fmax(d, min_lm_diagonal), max_lm_diagonal);
            D2[j] = d / radius;                       /* lm_diagonal = sqrt(diagonal/radius); D^2 enters the normal eqs */
        }
        int step_ok;
        double model_cost_change = 0.0;
        double *qr_buf = NULL;
        if (op->qr && jf) {
            /* DENSE_QR: [J S; D] y = [r; 0] in the least-squares sense, step = -y */
            const int m = nres + n;
            qr_buf = (double *)malloc(sizeof(double) * ((size_t)nres * n + (size_t)nres + (size_t)m * n + (size_t)m));
            double *Jm = qr_buf, *rv = Jm + (size_t)nres * n, *Aq = rv + nres, *bq = Aq + (size_t)m * n;
            jf(ctx, x, rv, Jm);
            for (int i = 0; i < nres; ++i) { for (int j = 0; j < n; ++j) { Jm[(size_t)n * i + j] *= scale[j]; Aq[(size_t)n * i + j] = Jm[(size_t)n * i + j]; } bq[i] = rv[i]; }
            for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) Aq[(size_t)n * (nres + i) + j] = (i == j) ? sqrt(D2[j]) : 0.0; bq[nres + i] = 0.0; }
            step_ok = orc_householder_ls(m, n, Aq, bq, y);
            if (step_ok) for (int j = 0; j < n; ++j) { step[j] = -y[j]; if (!isfinite(step[j])) step_ok = 0; }
            if (step_ok) {
                /* model_residuals = (J S) step ; model_cost_change = -model_residuals . (residuals + model_residuals / 2) */
                double acc = 0.0;
                for (int i = 0; i < nres; ++i) { double mr = 0.0; for (int j = 0; j < n; ++j) mr += Jm[(size_t)n * i + j] * step[j]; acc += mr * (rv[i] + 0.5 * mr); }
                model_cost_change = -acc;
                step_ok = (model_cost_change > 0.0);
            }
        } else {
            memcpy(A, Hs, sizeof(double) * n * n); for (int j = 0; j < n; ++j) A[(n + 1) * j] += D2[j];
            step_ok = orc_chol_solve(n, A, gs, y);
            if (step_ok) for (int j = 0; j < n; ++j) { step[j] = -y[j]; if (!isfinite(step[j])) step_ok = 0; }
            if (step_ok) {
                /* model_cost_change = -(J s)^T (r + J s / 2) = -(s^T gs + 1/2 s^T Hs s) */
                double sg = 0.0, sHs = 0.0;
                for (int a = 0; a < n; ++a) { sg += step[a] * gs[a]; double t = 0.0; for (int b = 0; b < n; ++b) t += Hs[n * a + b] * step[b]; sHs += step[a] * t; }
                model_cost_change = -(sg + 0.5 * sHs);
                step_ok = (model_cost_change > 0.0);
            }
        }
        free(qr_buf);
#define ORC_TRACE(cc, rd, sn, oc) do { if (op->trace && op->trace_n < op->trace_cap) { double *tr_ = op->trace + (size_t)ORC_TRACE_W * op->trace_n++; \
            tr_[0] = iteration; tr_[1] = trace_cost; tr_[2] = (cc); tr_[3] = model_cost_change; tr_[4] = (rd); tr_[5] = radius; tr_[6] = (sn); tr_[7] = (oc); } } while (0)
        if (!step_ok) {                                  /* HandleInvalidStep */
            if (++invalid_run >= max_consecutive_invalid) { ORC_TRACE(NAN, NAN, NAN, -2.0); sm->termination = ORC_FAILURE; sm->why = ORC_WHY_INVALID; sm->num_iterations = iteration; return; }
            radius *= 0.5;                               /* StepIsInvalid */
            ORC_TRACE(NAN, NAN, NAN, -1.0);
            continue;
        }
        invalid_run = 0;
        double delta[ORC_MAXN], cand[ORC_MAXN];
        for (int j = 0; j < n; ++j) { delta[j] = step[j] * scale[j]; cand[j] = x[j] + delta[j]; }
        double cand_cost, cg[ORC_MAXN], cH[ORC_MAXN * ORC_MAXN];
        if (!ev(ctx, cand, &cand_cost, cg, cH)) cand_cost = DBL_MAX;   /* ComputeCandidatePointAndEvaluateCost */
        /* ParameterToleranceReached — uses ||x - candidate|| */
        double step_norm = 0.0; for (int j = 0; j < n; ++j) { double d = x[j] - cand[j]; step_norm += d * d; } step_norm = sqrt(step_norm);
        if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) {
            ORC_TRACE(cand_cost, NAN, step_norm, 2.0);
            sm->termination = ORC_CONVERGENCE; sm->why = ORC_WHY_PARAMETER; sm->num_iterations = iteration; return; }
        /* FunctionToleranceReached */
        double cost_change = cost - cand_cost;
        if (fabs(cost_change) <= function_tolerance * cost) {
            ORC_TRACE(cand_cost, NAN, step_norm, 3.0);
            sm->termination = ORC_CONVERGENCE; sm->why = ORC_WHY_FUNCTION; sm->num_iterations = iteration; return; }
        double relative_decrease = cost_change / model_cost_change;        /* monotonic StepQuality */
        if (relative_decrease > min_relative_decrease) {                   /* HandleSuccessfulStep */
            memcpy(x, cand, sizeof(double) * n); x_norm = 0.0; for (int j = 0; j < n; ++j) x_norm += x[j] * x[j]; x_norm = sqrt(x_norm);
            cost = cand_cost; memcpy(g, cg, sizeof(double) * n); memcpy(H, cH, sizeof(double) * n * n);
            last_successful = 1; ++sm->num_successful;
            double t = 2.0 * relative_decrease - 1.0;                      /* StepAccepted */
            radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            radius = fmin(max_radius, radius);
            decrease_factor = 2.0;
            ORC_TRACE(cand_cost, relative_decrease, step_norm, 1.0);
        } else {                                                           /* HandleUnsuccessfulStep / StepRejected */
            radius = radius / decrease_factor; decrease_factor *= 2.0;
            ORC_TRACE(cand_cost, relative_decrease, step_norm, 0.0);
        }
    }
}

static void orc_lm_n(int n, orc_eval_fn ev, const void *ctx, const double *init, double *out, orc_lm_summary *sm) {
    orc_lm_opts op = orc_default_opts; op.qr = 0; op.trace = NULL;      /* the 7-parameter variants: normal equations only */
    orc_lm_n_ex(n, ev, NULL, 0, ctx, init, out, sm, &op);
}
static int orc_eval4_cb(const void *ctx, const double *x, double *cost, double *g, double *H) {
    return orc_eval((const orc_problem *)ctx, x, cost, g, H);
}
static void orc_jac4_cb(const void *ctx, const double *x, double *r, double *J) {
    const orc_problem *pb = (const orc_problem *)ctx;
    for (int i = 0; i < pb->pn; ++i)
        orc_residual_jet(&pb->cam, x, pb->pts2d[2 * i], pb->pts2d[2 * i + 1], pb->pts3d[3 * i], pb->pts3d[3 * i + 1],
                         pb->pts3d[3 * i + 2], pb->wgt2d[2 * i], pb->wgt2d[2 * i + 1], r + 2 * i, J + 8 * i);
}
static void orc_lm_opt(const orc_problem *pb, const double init[4], double out[4], orc_lm_summary *sm, orc_lm_opts *op) {
    orc_lm_n_ex(4, orc_eval4_cb, orc_jac4_cb, 2 * pb->pn, pb, init, out, sm, op);
}
static void orc_lm(const orc_problem *pb, const double init[4], double out[4], orc_lm_summary *sm) {
    orc_lm_opts op = orc_default_opts; op.trace = NULL;
    orc_lm_opt(pb, init, out, sm, &op);
}

/* ------------------------------------------------------------------------------------------
 * The reference's C entry point, same signature (src/ext.h:1-13, pnp_uncert_cpu.cpp:245-292).
 * result_cov (optional) = Ceres Covariance of the pose block = (J^T J)^-1 with the *Ceres*
 * Jacobian at the solution; failure (rank deficient) -> *result_val = 0, result_cov untouched.
 * ---------------------------------------------------------------------------------------- */
void orc_pnp_uncert_diag(double *pts2d, double *pts3d, double *wgt2d, double *K, double *init_pose,
                         int *result_val, double *result_pose, double *result_cov, double *result_tr,
                         int pn, double *clips, double *diag /* 6: iters, why, termination, init_cost, final_cost, n_success */) {
    orc_problem pb;
    pb.cam.fx = K[0]; pb.cam.fy = K[4]; pb.cam.cx = K[2]; pb.cam.cy = K[5];
    pb.cam.z_min = clips[0]; pb.cam.u_min = clips[1]; pb.cam.u_max = clips[2]; pb.cam.v_min = clips[3]; pb.cam.v_max = clips[4];
    pb.pn = pn; pb.pts2d = pts2d; pb.pts3d = pts3d; pb.wgt2d = wgt2d;
    orc_lm_summary sm;
    orc_lm(&pb, init_pose, result_pose, &sm);
    *result_val = (sm.termination == ORC_CONVERGENCE || sm.termination == ORC_NO_CONVERGENCE) ? 1 : 0;
    *result_tr = sm.radius;
    if (diag) { diag[0] = sm.num_iterations; diag[1] = sm.why; diag[2] = sm.termination; diag[3] = sm.initial_cost; diag[4] = sm.final_cost; diag[5] = sm.num_successful; }
    if (*result_val && result_cov) {
        double cost, g[4], H[16], inv[16];
        int ok = orc_eval(&pb, result_pose, &cost, g, H) && orc_spd_inverse4(H, inv);
        *result_val = ok ? 1 : 0;
        if (ok) memcpy(result_cov, inv, sizeof inv);
    }
}

/* the same solve with explicit options and an optional per-pass trace (trace_cap rows of ORC_TRACE_W doubles) */
int orc_pnp_uncert_opt(double *pts2d, double *pts3d, double *wgt2d, double *K, double *init_pose,
                       int *result_val, double *result_pose, double *result_tr, int pn, double *clips, double *diag,
                       int qr, int max_iter, double *trace, int trace_cap) {
    orc_problem pb;
    pb.cam.fx = K[0]; pb.cam.fy = K[4]; pb.cam.cx = K[2]; pb.cam.cy = K[5];
    pb.cam.z_min = clips[0]; pb.cam.u_min = clips[1]; pb.cam.u_max = clips[2]; pb.cam.v_min = clips[3]; pb.cam.v_max = clips[4];
    pb.pn = pn; pb.pts2d = pts2d; pb.pts3d = pts3d; pb.wgt2d = wgt2d;
    orc_lm_summary sm;
    orc_lm_opts op = { qr, max_iter > 0 ? max_iter : 50, trace, trace_cap, 0 };
    orc_lm_opt(&pb, init_pose, result_pose, &sm, &op);
    *result_val = (sm.termination == ORC_CONVERGENCE || sm.termination == ORC_NO_CONVERGENCE) ? 1 : 0;
    *result_tr = sm.radius;
    if (diag) { diag[0] = sm.num_iterations; diag[1] = sm.why; diag[2] = sm.termination; diag[3] = sm.initial_cost; diag[4] = sm.final_cost; diag[5] = sm.num_successful; }
    return op.trace_n;
}

void orc_pnp_uncert(double *pts2d, double *pts3d, double *wgt2d, double *K, double *init_pose,
                    int *result_val, double *result_pose, double *result_cov, double *result_tr,
                    int pn, double *clips) {
    orc_pnp_uncert_diag(pts2d, pts3d, wgt2d, K, init_pose, result_val, result_pose, result_cov, result_tr, pn, clips, NULL);
}


/* ------------------------------------------------------------------------------------------
 * N4: the two exported-but-never-called 7-parameter variants (pnp_uncert_cpu.cpp:294-377, ext.h:15-43).
 * dimpose = [log l, log h, log w, yaw, tx, ty, tz]; NocReprojectionErrorArray (:98-160) scales the NOC
 * point by exp(log-dims) before the same projection / clamps; NocCovReprojectionErrorArray (:163-228)
 * mixes the two residuals with a full 2x2 weight [wxx wxy; wxy wyy]; DimErrorArray (:77-95) is the prior
 * on the log-dims.  EVERY residual block (the 2-vectors and the 3-vector prior) goes through ONE
 * ceres::HuberLoss(delta) (:311,:324): restated with Ceres' Corrector (loss_function.h / corrector.cc:
 * rho'' <= 0 for Huber => residual and Jacobian are scaled by sqrt(rho'), block cost = rho(s)/2).
 * ---------------------------------------------------------------------------------------- */
static void orc_noc_residual_jet(const orc_cam *c, const double dp[7], double x2d, double y2d,
                                 double x3d, double y3d, double z3d, const double *w /*2 or 3*/, int full_cov,
                                 double res[2], double jac[14]) {
    jet7 p[7];
    for (int i = 0; i < 7; ++i) p[i] = j7_var(dp[i], i);
    jet7 pts3d[3] = { j7_mul(j7_const(x3d), j7_exp(p[0])), j7_mul(j7_const(y3d), j7_exp(p[1])), j7_mul(j7_const(z3d), j7_exp(p[2])) };
    jet7 r_vec[3] = { j7_const(0.0), p[3], j7_const(0.0) };
    jet7 t[3];
    j7_angle_axis_rotate_point(r_vec, pts3d, t);
    t[0] = j7_add(t[0], p[4]); t[1] = j7_add(t[1], p[5]); t[2] = j7_add(t[2], p[6]);
    if (t[2].a < c->z_min) t[2] = j7_const(c->z_min);
    jet7 proj_x = j7_add(j7_div(j7_mul(j7_const(c->fx), t[0]), t[2]), j7_const(c->cx));
    jet7 proj_y = j7_add(j7_div(j7_mul(j7_const(c->fy), t[1]), t[2]), j7_const(c->cy));
    if (proj_x.a < c->u_min) proj_x = j7_const(c->u_min); else if (proj_x.a > c->u_max) proj_x = j7_const(c->u_max);
    if (proj_y.a < c->v_min) proj_y = j7_const(c->v_min); else if (proj_y.a > c->v_max) proj_y = j7_const(c->v_max);
    jet7 dx = j7_sub(proj_x, j7_const(x2d)), dy = j7_sub(proj_y, j7_const(y2d));
    jet7 r0, r1;
    if (full_cov) {
        r0 = j7_add(j7_mul(j7_const(w[0]), dx), j7_mul(j7_const(w[1]), dy));
        r1 = j7_add(j7_mul(j7_const(w[1]), dx), j7_mul(j7_const(w[2]), dy));
    } else {
        r0 = j7_mul(j7_const(w[0]), dx);
        r1 = j7_mul(j7_const(w[1]), dy);
    }
    res[0] = r0.a; res[1] = r1.a;
    for (int i = 0; i < 7; ++i) { jac[i] = r0.v[i]; jac[7 + i] = r1.v[i]; }
}

/* ceres::HuberLoss::Evaluate + Corrector for one residual block of `nr` residuals (rows of J: 7 columns) */
static double orc_huber_correct(double delta, int nr, double *r, double *J) {
    double s = 0.0; for (int i = 0; i < nr; ++i) s += r[i] * r[i];
    const double b = delta * delta;
    double rho0, rho1;
    if (s > b) { const double q = sqrt(s); rho0 = 2.0 * delta * q - b; rho1 = fmax(DBL_MIN, delta / q); }
    else { rho0 = s; rho1 = 1.0; }
    const double sq = sqrt(rho1);                    /* rho'' <= 0 -> alpha = 0: plain scaling (corrector.cc) */
    for (int i = 0; i < nr; ++i) { r[i] *= sq; for (int j = 0; j < 7; ++j) J[7 * i + j] *= sq; }
    return 0.5 * rho0;
}

typedef struct {
    orc_cam cam; int pn, full_cov; double delta;
    const double *pts2d, *pts3d, *wgt2d, *logdim, *logdim_wgt;
} orc_noc_problem;

static int orc_noc_eval(const void *ctx, const double *x, double *cost, double *g, double *H) {
    const orc_noc_problem *pb = (const orc_noc_problem *)ctx;
    double c = 0.0, gg[7], HH[49];
    memset(gg, 0, sizeof gg); memset(HH, 0, sizeof HH);
    const int ws = pb->full_cov ? 3 : 2;
    for (int i = 0; i <= pb->pn; ++i) {
        double r[3], J[21]; int nr;
        if (i < pb->pn) {
            nr = 2;
            orc_noc_residual_jet(&pb->cam, x, pb->pts2d[2 * i], pb->pts2d[2 * i + 1], pb->pts3d[3 * i], pb->pts3d[3 * i + 1],
                                 pb->pts3d[3 * i + 2], pb->wgt2d + ws * i, pb->full_cov, r, J);
        } else {                                     /* DimErrorArray: w_k (dimpose[k] - logdim[k]) */
            nr = 3; memset(J, 0, sizeof J);
            for (int k = 0; k < 3; ++k) { r[k] = pb->logdim_wgt[k] * (x[k] - pb->logdim[k]); J[7 * k + k] = pb->logdim_wgt[k]; }
        }
        c += orc_huber_correct(pb->delta, nr, r, J);
        for (int a = 0; a < 7; ++a) {
            for (int q = 0; q < nr; ++q) gg[a] += J[7 * q + a] * r[q];
            for (int b2 = 0; b2 < 7; ++b2) for (int q = 0; q < nr; ++q) HH[7 * a + b2] += J[7 * q + a] * J[7 * q + b2];
        }
    }
    *cost = c;
    int ok = isfinite(c);
    for (int a = 0; a < 7; ++a) { g[a] = gg[a]; ok = ok && isfinite(gg[a]); }
    for (int a = 0; a < 49; ++a) { H[a] = HH[a]; ok = ok && isfinite(HH[a]); }
    return ok;
}

static void orc_noc_solve(int full_cov, double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt,
                          double *K, double *init_dimpose, int *result_val, double *result_dimpose, int pn, double *clips,
                          double delta, double *diag) {
    orc_noc_problem pb;
    pb.cam.fx = K[0]; pb.cam.fy = K[4]; pb.cam.cx = K[2]; pb.cam.cy = K[5];
    pb.cam.z_min = clips[0]; pb.cam.u_min = clips[1]; pb.cam.u_max = clips[2]; pb.cam.v_min = clips[3]; pb.cam.v_max = clips[4];
    pb.pn = pn; pb.full_cov = full_cov; pb.delta = delta;
    pb.pts2d = pts2d; pb.pts3d = pts3d; pb.wgt2d = wgt2d; pb.logdim = logdim; pb.logdim_wgt = logdim_wgt;
    orc_lm_summary sm;
    orc_lm_n(7, orc_noc_eval, &pb, init_dimpose, result_dimpose, &sm);
    *result_val = (sm.termination == ORC_CONVERGENCE || sm.termination == ORC_NO_CONVERGENCE) ? 1 : 0;
    if (diag) { diag[0] = sm.num_iterations; diag[1] = sm.why; diag[2] = sm.termination; diag[3] = sm.initial_cost; diag[4] = sm.final_cost; diag[5] = sm.num_successful; }
}

void orc_pnp_noc_uncert(double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt, double *K,
                        double *init_dimpose, int *result_val, double *result_dimpose, int pn, double *clips, double delta) {
    orc_noc_solve(0, pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, init_dimpose, result_val, result_dimpose, pn, clips, delta, NULL);
}
void orc_pnp_noc_cov_uncert(double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt, double *K,
                            double *init_dimpose, int *result_val, double *result_dimpose, int pn, double *clips, double delta) {
    orc_noc_solve(1, pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, init_dimpose, result_val, result_dimpose, pn, clips, delta, NULL);
}
/* test hooks: cost / gradient / J^T J of the robustified problem at x, and the solve with diagnostics */
int orc_noc_cost_grad(int full_cov, double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt, double *K,
                      double *x, int pn, double *clips, double delta, double *cost, double *g, double *H) {
    orc_noc_problem pb;
    pb.cam.fx = K[0]; pb.cam.fy = K[4]; pb.cam.cx = K[2]; pb.cam.cy = K[5];
    pb.cam.z_min = clips[0]; pb.cam.u_min = clips[1]; pb.cam.u_max = clips[2]; pb.cam.v_min = clips[3]; pb.cam.v_max = clips[4];
    pb.pn = pn; pb.full_cov = full_cov; pb.delta = delta;
    pb.pts2d = pts2d; pb.pts3d = pts3d; pb.wgt2d = wgt2d; pb.logdim = logdim; pb.logdim_wgt = logdim_wgt;
    return orc_noc_eval(&pb, x, cost, g, H);
}
void orc_noc_solve_diag(int full_cov, double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt, double *K,
                        double *init_dimpose, int *result_val, double *result_dimpose, int pn, double *clips, double delta, double *diag) {
    orc_noc_solve(full_cov, pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, init_dimpose, result_val, result_dimpose, pn, clips, delta, diag);
}

/* ------------------------------------------------------------------------------------------
 * R2/R7: torch-semantics Jacobian and approx Hessian (jacobian.py:4-98, hessian.py:67-87), one object.
 * Full 2x3 upper K is honoured (forward_proj multiplies by cam_mats, :20-26); depth = 3rd row of
 * K R X + K t.  zero_mask = z_clip | uv_clip(per axis) | outlier  (jacobian.py:52-59).
 * Column order [yaw, tx, ty, tz]; rows point-major [u0, v0, u1, v1, ...].
 * jac (pn,2,4) and err (pn,2) may be NULL.  All fp64.
 * ---------------------------------------------------------------------------------------- */
void orc_torch_jacobian(const double *K /*9*/, double z_min, const double *u_range, const double *v_range,
                        double yaw, const double *t, const double *pts2d, const double *pts3d,
                        const double *istd, const uint8_t *inlier /*nullable*/, int pn,
                        double *jac, double *err, double *H /*16*/) {
    const double s = sin(yaw), c = cos(yaw);
    /* k_r = K * R_y,  R_y = [[c,0,s],[0,1,0],[-s,0,c]] ;  k_t = K t */
    double kr[9], kt[3];
    for (int r = 0; r < 3; ++r) {
        kr[3 * r + 0] = K[3 * r + 0] * c - K[3 * r + 2] * s;
        kr[3 * r + 1] = K[3 * r + 1];
        kr[3 * r + 2] = K[3 * r + 0] * s + K[3 * r + 2] * c;
        kt[r] = K[3 * r + 0] * t[0] + K[3 * r + 1] * t[1] + K[3 * r + 2] * t[2];
    }
    /* jac_yaw_m1 = K[0:2,[0,2]] * [[-s, c],[-c,-s]]  (jacobian.py:74-81) */
    const double m1[4] = { K[0] * (-s) + K[2] * (-c), K[0] * c + K[2] * (-s),
                           K[3] * (-s) + K[5] * (-c), K[3] * c + K[5] * (-s) };
    double HH[16]; memset(HH, 0, sizeof HH);
    for (int i = 0; i < pn; ++i) {
        const double X = pts3d[3 * i], Y = pts3d[3 * i + 1], Z = pts3d[3 * i + 2];
        double uvz[3];
        for (int r = 0; r < 3; ++r) uvz[r] = kr[3 * r] * X + kr[3 * r + 1] * Y + kr[3 * r + 2] * Z + kt[r];
        double z = uvz[2]; const int zclip = z < z_min; if (zclip) z = z_min;
        double uv[2] = { uvz[0] / z, uvz[1] / z };
        const double lb[2] = { u_range[0], v_range[0] }, ub[2] = { u_range[1], v_range[1] };
        int clip[2];
        for (int a = 0; a < 2; ++a) { clip[a] = (uv[a] < lb[a]) || (uv[a] > ub[a]); uv[a] = fmax(lb[a], fmin(ub[a], uv[a])); }
        const int outl = inlier ? !inlier[i] : 0;
        double J[8];
        for (int a = 0; a < 2; ++a) {
            const double w = istd[2 * i + a];
            const int zero = zclip || clip[a] || outl;
            /* jac_t_vec = [K[a,0], K[a,1], K[a,2]-uv[a]] / z * istd */
            double jt0 = K[3 * a + 0] / z, jt1 = K[3 * a + 1] / z, jt2 = (K[3 * a + 2] - uv[a]) / z;
            /* jac_yaw = ((m1[a,:] + uv[a]*[c,s]) . [X, Z]) / z * istd */
            double jy = ((m1[2 * a] + uv[a] * c) * X + (m1[2 * a + 1] + uv[a] * s) * Z) / z;
            J[4 * a + 0] = zero ? 0.0 : jy * w;
            J[4 * a + 1] = zero ? 0.0 : jt0 * w;
            J[4 * a + 2] = zero ? 0.0 : jt1 * w;
            J[4 * a + 3] = zero ? 0.0 : jt2 * w;
            if (err) err[2 * i + a] = (uv[a] - pts2d[2 * i + a]) * w;   /* error is NOT masked (jacobian.py:163-165) */
        }
        if (jac) memcpy(jac + 8 * i, J, sizeof J);
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) HH[4 * a + b] += J[a] * J[b] + J[4 + a] * J[4 + b];
    }
    if (H) memcpy(H, HH, sizeof HH);
}

/* ------------------------------------------------------------------------------------------
 * The covariance Hessian AS THE KERNEL'S STAGE 4 SPECIFIES IT (round 6; csrc/pnp_kernel_body.inc "stage 4", operation for operation):
 * the same J^T J as orc_torch_jacobian up to rounding — the reference's own is a BLAS product of unspecified order (hessian.py:85-86), so
 * neither is "the" order — but with every operation and the summation tree fixed, so that a Hessian that is singular to rounding
 * (planar / collinear object points) factorises, or fails to, identically on both sides and `valid` can be compared bit for bit:
 *   - sin / cos of the float32 yaw: orc_spec_sincos (Cody-Waite reduction + the minimax kernels, every fma written);
 *   - no contraction anywhere else (oracle/Makefile: -ffp-contract=off), one IEEE division per point (iz = 1 / z), w = istd * iz;
 *   - only the inliers contribute (outlier rows are ASSIGNED zero, jacobian.py:52-59); thread t of the object's 64 x waves threads
 *     accumulates inliers t, t + 64 waves, ... in ascending order (q-th inlier = q-th set bit of the mask) with
 *     acc = fma(Ju_i, Ju_j, fma(Jv_i, Jv_j, acc)); the 64 partials of a wave combine by the butterfly with strides 32, 16, 1, 2, 4, 8
 *     (v_permlane32_swap, v_permlane16_swap, four DPP steps: epnp_tree64), the waves' totals add up in wave order.
 * waves = the kernel's waves per object (4 for launches of fewer than 2048 objects, 2 beyond; mr_pick_waves), 1..8.
 * ---------------------------------------------------------------------------------------- */
static void orc_spec_sincos(double x, double *sn, double *cs) {
    const double kd = rint(x * 6.36619772367581382433e-01);
    double r = fma(-kd, 1.57079632673412561417e+00, x);
    r = fma(-kd, 6.07710050630396597660e-11, r);
    r = fma(-kd, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double zr = z * r;
    const double s = fma(zr, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double t1 = 1.0 - w, t2 = t1 - hz, zz = z * z;
    const double c = w + fma(zz, pc, t2);
    /* (int)kd as v_cvt_i32_f64 converts: saturating, NaN -> 0 */
    const int qi = (kd != kd) ? 0 : (kd >= 2147483647.0 ? 2147483647 : (kd <= -2147483648.0 ? (-2147483647 - 1) : (int)kd));
    const int q = qi & 3;
    const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}
void orc_spec_sincos_export(double x, double *sn, double *cs) { orc_spec_sincos(x, sn, cs); }

static double orc_tree64(double *p /* [64], destroyed */) {            /* = epnp_tree64 (epnp.inc is included further down) */
    static const int stride[6] = { 32, 16, 1, 2, 4, 8 };
    double q[64];
    for (int t = 0; t < 6; ++t) { for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ stride[t]]; memcpy(p, q, sizeof q); }
    return p[0];
}

void orc_cov_hessian_spec(const double *K /*9*/, double z_min, const double *u_range, const double *v_range,
                          double yaw, const double *t, const double *pts3d, const double *istd, const uint8_t *inlier /*nullable: all*/,
                          int pn, int waves, double *H /*16*/) {
    double sn, cs;
    orc_spec_sincos(yaw, &sn, &cs);
    const double tx = t[0], ty = t[1], tz = t[2];
    double kr[9], kt[3];
    for (int r = 0; r < 3; ++r) {
        kr[3 * r + 0] = K[3 * r + 0] * cs - K[3 * r + 2] * sn;
        kr[3 * r + 1] = K[3 * r + 1];
        kr[3 * r + 2] = K[3 * r + 0] * sn + K[3 * r + 2] * cs;
        kt[r] = K[3 * r + 0] * tx + K[3 * r + 1] * ty + K[3 * r + 2] * tz;
    }
    const double m1[4] = { K[0] * (-sn) + K[2] * (-cs), K[0] * cs + K[2] * (-sn),
                           K[3] * (-sn) + K[5] * (-cs), K[3] * cs + K[5] * (-sn) };
    if (waves < 1) waves = 1;
    if (waves > 8) waves = 8;
    const int NT = 64 * waves;
    double (*part)[10] = (double (*)[10])calloc((size_t)NT, sizeof(double[10]));
    int qi = 0;
    for (int i = 0; i < pn; ++i) {
        if (inlier && !inlier[i]) continue;
        double *hacc = part[qi % NT];
        ++qi;
        const double X = pts3d[3 * i], Y = pts3d[3 * i + 1], Z = pts3d[3 * i + 2];
        const double un = kr[0] * X + kr[1] * Y + kr[2] * Z + kt[0];
        const double vn = kr[3] * X + kr[4] * Y + kr[5] * Z + kt[1];
        double z = kr[6] * X + kr[7] * Y + kr[8] * Z + kt[2];
        const int zclip = z < z_min;
        z = zclip ? z_min : z;
        const double iz = 1.0 / z;
        double uv[2] = { un * iz, vn * iz };
        const int cl[2] = { (uv[0] < u_range[0]) || (uv[0] > u_range[1]), (uv[1] < v_range[0]) || (uv[1] > v_range[1]) };
        uv[0] = fmax(u_range[0], fmin(u_range[1], uv[0]));
        uv[1] = fmax(v_range[0], fmin(v_range[1], uv[1]));
        double J[8];
        for (int r = 0; r < 2; ++r) {
            const int zero = zclip || cl[r];
            const double w = istd[2 * i + r] * iz;
            J[4 * r + 0] = zero ? 0.0 : w * ((m1[2 * r] + uv[r] * cs) * X + (m1[2 * r + 1] + uv[r] * sn) * Z);
            J[4 * r + 1] = zero ? 0.0 : w * K[3 * r + 0];
            J[4 * r + 2] = zero ? 0.0 : w * K[3 * r + 1];
            J[4 * r + 3] = zero ? 0.0 : w * (K[3 * r + 2] - uv[r]);
        }
        int q = 0;
        for (int a = 0; a < 4; ++a) for (int b = a; b < 4; ++b, ++q) hacc[q] = fma(J[a], J[b], fma(J[4 + a], J[4 + b], hacc[q]));
    }
    double tot[10];
    for (int q = 0; q < 10; ++q) {
        double acc = 0.0;
        for (int w = 0; w < waves; ++w) {
            double p64[64];
            for (int l = 0; l < 64; ++l) p64[l] = part[64 * w + l][q];
            const double ws = orc_tree64(p64);
            acc = (w == 0) ? ws : acc + ws;
        }
        tot[q] = acc;
    }
    free(part);
    int q = 0;
    for (int a = 0; a < 4; ++a) for (int b = a; b < 4; ++b, ++q) H[4 * a + b] = H[4 * b + a] = tot[q];
}

/* ------------------------------------------------------------------------------------------
 * exact_hessian (hessian.py:5-64): h[i][j] = d/d pose_j of (J^T e)_i, which the reference obtains by torch autograd
 * through the ANALYTIC Jacobian expressions of get_pose_jacobians (jacobian.py:48-98) and the weighted error of
 * forward_proj (:4-45); the masks (z clip, per-axis uv clip, outliers) are constants for autograd and masked rows were
 * ASSIGNED zero, so only unmasked rows contribute, and on those every expression is smooth.  Restated in closed form,
 * differentiating exactly the reference's expressions (general K, including a third row that is not (0,0,1)):
 *     nu_r = k_r[r].X + k_t[r],  z = k_r[2].X + k_t[2],  uv_r = nu_r / z,  Bv = c X + s Z,  A = -s X + c Z
 *     J_r,i = w_r a_i / z,   a = [ m1[r,0] X + m1[r,1] Z + uv_r Bv,  K[r,0],  K[r,1],  K[r,2] - uv_r ]
 *     D_j nu_r = [K[r,0] A - K[r,2] Bv, K[r,0], K[r,1], K[r,2]],  D_j z likewise with r = 2,  D_j uv_r = (D_j nu_r - uv_r D_j z) / z
 *     D_j a_0 = D_j uv_r Bv + [j = yaw] (-K[r,0] Bv - K[r,2] A + uv_r A),   D_j a_3 = -D_j uv_r
 *     h[i][j] = sum_r  e_r w_r (D_j a_i / z - a_i D_j z / z^2)  +  (w_r a_i / z) (w_r D_j uv_r),     e_r = w_r (uv_r - x2d_r)
 * For K with third row (0,0,1) this is J^T J + sum_r e_r w_r Hess(uv_r), symmetric.  All fp64.
 * ---------------------------------------------------------------------------------------- */
void orc_exact_hessian(const double *K /*9*/, double z_min, const double *u_range, const double *v_range,
                       double yaw, const double *t, const double *pts2d, const double *pts3d,
                       const double *istd, const uint8_t *inlier /*nullable*/, int pn, double *H /*16*/) {
    const double s = sin(yaw), c = cos(yaw);
    double kr[9], kt[3];
    for (int r = 0; r < 3; ++r) {
        kr[3 * r + 0] = K[3 * r + 0] * c - K[3 * r + 2] * s;
        kr[3 * r + 1] = K[3 * r + 1];
        kr[3 * r + 2] = K[3 * r + 0] * s + K[3 * r + 2] * c;
        kt[r] = K[3 * r + 0] * t[0] + K[3 * r + 1] * t[1] + K[3 * r + 2] * t[2];
    }
    const double m1[4] = { K[0] * (-s) + K[2] * (-c), K[0] * c + K[2] * (-s),
                           K[3] * (-s) + K[5] * (-c), K[3] * c + K[5] * (-s) };
    double HH[16]; memset(HH, 0, sizeof HH);
    for (int i = 0; i < pn; ++i) {
        if (inlier && !inlier[i]) continue;
        const double X = pts3d[3 * i], Y = pts3d[3 * i + 1], Z = pts3d[3 * i + 2];
        double uvz[3];
        for (int r = 0; r < 3; ++r) uvz[r] = kr[3 * r] * X + kr[3 * r + 1] * Y + kr[3 * r + 2] * Z + kt[r];
        const double z = uvz[2];
        if (z < z_min) continue;                                   /* z clip masks both rows */
        const double Bv = c * X + s * Z, A = -s * X + c * Z;
        const double dz[4] = { K[6] * A - K[8] * Bv, K[6], K[7], K[8] };
        const double lb[2] = { u_range[0], v_range[0] }, ub[2] = { u_range[1], v_range[1] };
        for (int r = 0; r < 2; ++r) {
            const double uv = uvz[r] / z;
            if (uv < lb[r] || uv > ub[r]) continue;                /* per-axis clip masks this row */
            const double w = istd[2 * i + r], e = w * (uv - pts2d[2 * i + r]);
            const double dn[4] = { K[3 * r] * A - K[3 * r + 2] * Bv, K[3 * r], K[3 * r + 1], K[3 * r + 2] };
            double du[4], a[4], Da[16];
            for (int j = 0; j < 4; ++j) du[j] = (dn[j] - uv * dz[j]) / z;
            a[0] = m1[2 * r] * X + m1[2 * r + 1] * Z + uv * Bv; a[1] = K[3 * r]; a[2] = K[3 * r + 1]; a[3] = K[3 * r + 2] - uv;
            memset(Da, 0, sizeof Da);
            for (int j = 0; j < 4; ++j) { Da[j] = du[j] * Bv; Da[12 + j] = -du[j]; }
            Da[0] += -K[3 * r] * Bv - K[3 * r + 2] * A + uv * A;
            for (int ii = 0; ii < 4; ++ii)
                for (int j = 0; j < 4; ++j)
                    HH[4 * ii + j] += e * w * (Da[4 * ii + j] / z - a[ii] * dz[j] / (z * z)) + (w * a[ii] / z) * (w * du[j]);
        }
    }
    memcpy(H, HH, sizeof HH);
}

/* general 4x4 inverse (Gauss-Jordan, partial pivoting: what torch.inverse's LU does, pnp_uncert.py:77-78); 0 when a pivot is
 * exactly zero or the result is not finite */
int orc_inverse4(const double H[16], double inv[16]) {
    double M[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { M[i][j] = H[4 * i + j]; M[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
        if (!(fabs(M[piv][col]) > 0.0)) return 0;
        if (piv != col) for (int j = 0; j < 8; ++j) { const double tmp = M[col][j]; M[col][j] = M[piv][j]; M[piv][j] = tmp; }
        const double d = 1.0 / M[col][col];
        for (int j = 0; j < 8; ++j) M[col][j] *= d;
        for (int r = 0; r < 4; ++r) if (r != col) { const double f = M[r][col]; for (int j = 0; j < 8; ++j) M[r][j] -= f * M[col][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { inv[4 * i + j] = M[i][4 + j]; if (!isfinite(inv[4 * i + j])) return 0; }
    return 1;
}

/* pose_cov for forward_exact_hessian=True: inverse(h) by LU (h need not be positive definite); singular -> identity + invalid */
int orc_pose_cov_general(const double H[16], double cov[16]) {
    if (orc_inverse4(H, cov)) return 1;
    for (int i = 0; i < 16; ++i) cov[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return 0;
}

/* R6: pose_cov = inverse(h); singular -> identity + invalid (the per-object reading of pnp_uncert.py:77-85) */
int orc_pose_cov(const double H[16], double cov[16]) {
    if (orc_spd_inverse4(H, cov)) return 1;
    for (int i = 0; i < 16; ++i) cov[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * K0: deterministic initialiser + consensus inlier selection (this repo's replacement for
 * cv2.solvePnPRansac(EPNP, 30 iters) / cv2.solvePnP(EPNP), pnp_uncert_cpu.py:35-58).  Spec in DESIGN.md §K0.
 *   - candidates: the istd-inlier set (mask0), listed in ascending point index
 *   - hypothesis h = 0..n_hyp-1: 5 points, one per fifth of the candidate list (stratified),
 *     picked by a counter-based hash; linear 4-DoF solve (5x5 normal eqs in (cos,sin,tx,ty,tz)),
 *     normalise (cos,sin), re-solve t (3x3) — fp64, fixed operation order, explicit fma
 *   - consensus of every hypothesis over the candidates in fp32 with a fixed operation order:
 *     |fx X - (u-cx) Z|^2 + |fy Y - (v-cy) Z|^2 <= (thr Z) |thr Z|   (false for Z < 0 and for NaN)
 *   - hypotheses are scored in blocks of 8 with RANSAC's adaptive stop (confidence 0.99);
 *     best = first maximum; fewer than 5 consensus points -> failure (ret False, like RANSAC)
 *   - refit on the consensus set with the same linear solver (fp64), yaw0 = atan2(sin, cos)
 * ---------------------------------------------------------------------------------------- */
#define ORC_K0_SEED 0x9E3779B9u
static uint32_t orc_hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

typedef struct { double fx, fy, cx, cy; } orc_k4;

/* accumulate one point into the 5x5 system; n5: upper triangle row-major (15), m5: rhs (5) */
static void orc_lin5_add(const orc_k4 *k, float u, float v, float x, float y, float z, double n5[15], double m5[5]) {
    const double a = (double)u - k->cx, b = (double)v - k->cy;
    const double X = x, Y = y, Z = z;
    const double ru[5] = { fma(a, Z, -(k->fx * X)), fma(-a, X, -(k->fx * Z)), -k->fx, 0.0, a };
    const double rv[5] = { b * Z, -(b * X), 0.0, -k->fy, b };
    const double rhs_v = k->fy * Y;
    int q = 0;
    for (int i = 0; i < 5; ++i) {
        for (int j = i; j < 5; ++j, ++q) n5[q] = fma(rv[i], rv[j], fma(ru[i], ru[j], n5[q]));
        m5[i] = fma(rv[i], rhs_v, m5[i]);
    }
}
/* accumulate one point into the 3x3 translation system given unit (c,s) */
static void orc_lin3_add(const orc_k4 *k, double c, double s, float u, float v, float x, float y, float z, double n3[4], double m3[3]) {
    const double a = (double)u - k->cx, b = (double)v - k->cy;
    const double X = x, Y = y, Z = z;
    const double Xr = fma(c, X, s * Z), Zr = fma(c, Z, -(s * X));
    const double bu = fma(k->fx, Xr, -(a * Zr));       /* -fx tx + a tz = fx Xr - a Zr */
    const double bv = fma(k->fy, Y, -(b * Zr));        /* -fy ty + b tz = fy Y  - b Zr */
    /* n3 = { sum fx^2 (count), sum a, sum b, sum a^2+b^2 } – the structurally non-zero sums */
    n3[0] += 1.0; n3[1] += a; n3[2] += b; n3[3] = fma(b, b, fma(a, a, n3[3]));
    m3[0] = fma(-k->fx, bu, m3[0]); m3[1] = fma(-k->fy, bv, m3[1]); m3[2] = fma(b, bv, fma(a, bu, m3[2]));
}
static int orc_lin5_solve(const double n5[15], const double m5[5], double *c, double *s) {
    double A[25], th[5]; int q = 0;
    for (int i = 0; i < 5; ++i) for (int j = i; j < 5; ++j, ++q) A[5 * i + j] = A[5 * j + i] = n5[q];
    if (!orc_chol_solve(5, A, m5, th)) return 0;
    const double rho2 = fma(th[0], th[0], th[1] * th[1]);
    if (!(rho2 > 1e-12) || !isfinite(rho2)) return 0;
    const double inv = 1.0 / sqrt(rho2);
    *c = th[0] * inv; *s = th[1] * inv;
    return 1;
}
static int orc_lin3_solve(const orc_k4 *k, const double n3[4], const double m3[3], double t[3]) {
    const double A[9] = { k->fx * k->fx * n3[0], 0.0, -k->fx * n3[1],
                          0.0, k->fy * k->fy * n3[0], -k->fy * n3[2],
                          -k->fx * n3[1], -k->fy * n3[2], n3[3] };
    if (!orc_chol_solve(3, A, m3, t)) return 0;
    return isfinite(t[0]) && isfinite(t[1]) && isfinite(t[2]);
}
static int orc_consensus(const float hyp[5], float fx, float fy, float a, float b, float x, float y, float z, float thr) {
    const float c = hyp[0], s = hyp[1];
    const float Xc = fmaf(c, x, fmaf(s, z, hyp[2]));
    const float Zc = fmaf(c, z, fmaf(-s, x, hyp[4]));
    const float Yc = y + hyp[3];
    const float eu = fmaf(-a, Zc, fx * Xc);
    const float ev = fmaf(-b, Zc, fy * Yc);
    const float e2 = fmaf(eu, eu, ev * ev);
    const float lim = thr * Zc;
    return e2 <= lim * fabsf(lim);            /* one compare: Zc <= 0 makes the bound <= 0, NaN compares false */
}

/* returns 1 on success.  mask: in = candidates (mask0), out = consensus set (if ransac) . */
int orc_k0_init(const float *x2d /*pn,2*/, const float *x3d /*pn,3*/, uint8_t *mask, int pn,
                const float *K /*9, f32*/, int use_ransac, float thr, int n_hyp,
                double init_pose[4], int *best_hyp, int *best_count) {
    const float fxf = K[0], fyf = K[4], cxf = K[2], cyf = K[5];
    const orc_k4 k = { (double)fxf, (double)fyf, (double)cxf, (double)cyf };
    int *list = (int *)malloc(sizeof(int) * (size_t)(pn > 0 ? pn : 1)); int n = 0;
    for (int p = 0; p < pn; ++p) if (mask[p]) list[n++] = p;
    if (best_hyp) *best_hyp = -1; if (best_count) *best_count = n;
    int ok = 1;
    if (use_ransac) {
        if (n_hyp > 64) n_hyp = 64;
        float hyp[64][5]; int valid[64];
        for (int h = 0; h < n_hyp; ++h) {
            valid[h] = 0;
            if (n < 5) continue;
            double n5[15], m5[5], n3[4], m3[3], c, s, t[3]; int idx[5];
            memset(n5, 0, sizeof n5); memset(m5, 0, sizeof m5); memset(n3, 0, sizeof n3); memset(m3, 0, sizeof m3);
            for (int j = 0; j < 5; ++j) {
                const uint32_t lo = (uint32_t)(((uint64_t)j * (uint64_t)n) / 5u), hi = (uint32_t)(((uint64_t)(j + 1) * (uint64_t)n) / 5u);
                const uint32_t r = lo + (uint32_t)(((uint64_t)orc_hash32(ORC_K0_SEED + (uint32_t)h * 8u + (uint32_t)j) * (uint64_t)(hi - lo)) >> 32);
                idx[j] = list[r];
                const int p = idx[j];
                orc_lin5_add(&k, x2d[2 * p], x2d[2 * p + 1], x3d[3 * p], x3d[3 * p + 1], x3d[3 * p + 2], n5, m5);
            }
            if (!orc_lin5_solve(n5, m5, &c, &s)) continue;
            for (int j = 0; j < 5; ++j) { const int p = idx[j];
                orc_lin3_add(&k, c, s, x2d[2 * p], x2d[2 * p + 1], x3d[3 * p], x3d[3 * p + 1], x3d[3 * p + 2], n3, m3); }
            if (!orc_lin3_solve(&k, n3, m3, t)) continue;
            hyp[h][0] = (float)c; hyp[h][1] = (float)s; hyp[h][2] = (float)t[0]; hyp[h][3] = (float)t[1]; hyp[h][4] = (float)t[2];
            valid[h] = 1;
        }
        /* consensus in blocks of 8 hypotheses with RANSAC's adaptive termination (confidence 0.99,
         * 5-point samples: stop once (1 - w^5)^evaluated <= 0.01, w = best inlier ratio) — the rule
         * cv2.solvePnPRansac applies after every iteration, here checked after every block. */
        int best = -1, bestc = 0;
        for (int h0 = 0; h0 < n_hyp; h0 += 8) {
            for (int h = h0; h < h0 + 8 && h < n_hyp; ++h) {
                if (!valid[h]) continue;
                int cnt = 0;
                for (int q = 0; q < n; ++q) { const int p = list[q];
                    cnt += orc_consensus(hyp[h], fxf, fyf, x2d[2 * p] - cxf, x2d[2 * p + 1] - cyf, x3d[3 * p], x3d[3 * p + 1], x3d[3 * p + 2], thr); }
                if (cnt > bestc) { bestc = cnt; best = h; }
            }
            if (bestc >= 5) {
                const double w = (double)bestc / (double)n;
                double w5 = w * w; w5 = w5 * w5 * w;
                double q8 = 1.0 - w5; q8 = q8 * q8; q8 = q8 * q8; q8 = q8 * q8;
                double qk = q8;
                for (int k = 8; k < h0 + 8; k += 8) qk = qk * q8;
                if (qk <= 0.01) break;
            }
        }
        if (best_hyp) *best_hyp = best; if (best_count) *best_count = bestc;
        if (best < 0 || bestc < 5) ok = 0;
        else {
            int m = 0;
            for (int q = 0; q < n; ++q) { const int p = list[q];
                const int in = orc_consensus(hyp[best], fxf, fyf, x2d[2 * p] - cxf, x2d[2 * p + 1] - cyf, x3d[3 * p], x3d[3 * p + 1], x3d[3 * p + 2], thr);
                mask[p] = (uint8_t)in; if (in) list[m++] = p; }
            n = m;
        }
    }
    if (ok) {   /* refit on the final set (the analogue of the final EPnP on the inliers) */
        double n5[15], m5[5], n3[4], m3[3], c, s, t[3];
        memset(n5, 0, sizeof n5); memset(m5, 0, sizeof m5); memset(n3, 0, sizeof n3); memset(m3, 0, sizeof m3);
        for (int q = 0; q < n; ++q) { const int p = list[q]; orc_lin5_add(&k, x2d[2 * p], x2d[2 * p + 1], x3d[3 * p], x3d[3 * p + 1], x3d[3 * p + 2], n5, m5); }
        ok = orc_lin5_solve(n5, m5, &c, &s);
        if (ok) { for (int q = 0; q < n; ++q) { const int p = list[q]; orc_lin3_add(&k, c, s, x2d[2 * p], x2d[2 * p + 1], x3d[3 * p], x3d[3 * p + 1], x3d[3 * p + 2], n3, m3); }
                  ok = orc_lin3_solve(&k, n3, m3, t); }
        if (ok) { init_pose[0] = atan2(s, c); init_pose[1] = t[0]; init_pose[2] = t[1]; init_pose[3] = t[2]; }
    }
    free(list);
    return ok;
}

/* ------------------------------------------------------------------------------------------
 * N4 remainder: a TRUE 6-DoF variant — what `use_6dof=True` would mean (pnp_uncert.py:11 accepts the flag and ignores it; the
 * north-star's "6x6 normal equations").  No reference code exists for it: the SAME residual functor as R1
 * (pnp_uncert_cpu.cpp:24-51) with the full angle-axis vector r = (rx, ry, rz) in ceres::AngleAxisRotatePoint instead of
 * (0, yaw, 0), pose6 = [rx, ry, rz, tx, ty, tz], Jets with 6 partials, the same Ceres-1.14 LM, covariance = (J^T J)^-1 with
 * the solver's Jacobian.  Pinned by finite differences and scipy (tests/test_pnp6.py).
 * ---------------------------------------------------------------------------------------- */
static void orc_residual_jet6(const orc_cam *c, const double pose[6], double x2d, double y2d, double x3d, double y3d, double z3d,
                              double wxx, double wyy, double res[2], double jac[12]) {
    jet6 p[6]; for (int i = 0; i < 6; ++i) p[i] = j6_var(pose[i], i);
    jet6 pts3d[3] = { j6_const(x3d), j6_const(y3d), j6_const(z3d) };
    jet6 r_vec[3] = { p[0], p[1], p[2] };
    jet6 t[3];
    j6_angle_axis_rotate_point(r_vec, pts3d, t);
    t[0] = j6_add(t[0], p[3]); t[1] = j6_add(t[1], p[4]); t[2] = j6_add(t[2], p[5]);
    if (t[2].a < c->z_min) t[2] = j6_const(c->z_min);
    jet6 proj_x = j6_add(j6_div(j6_mul(j6_const(c->fx), t[0]), t[2]), j6_const(c->cx));
    jet6 proj_y = j6_add(j6_div(j6_mul(j6_const(c->fy), t[1]), t[2]), j6_const(c->cy));
    if (proj_x.a < c->u_min) proj_x = j6_const(c->u_min); else if (proj_x.a > c->u_max) proj_x = j6_const(c->u_max);
    if (proj_y.a < c->v_min) proj_y = j6_const(c->v_min); else if (proj_y.a > c->v_max) proj_y = j6_const(c->v_max);
    jet6 r0 = j6_mul(j6_const(wxx), j6_sub(proj_x, j6_const(x2d)));
    jet6 r1 = j6_mul(j6_const(wyy), j6_sub(proj_y, j6_const(y2d)));
    res[0] = r0.a; res[1] = r1.a;
    for (int i = 0; i < 6; ++i) { jac[i] = r0.v[i]; jac[6 + i] = r1.v[i]; }
}
static int orc_eval6_cb(const void *ctx, const double *x, double *cost, double *g, double *H) {
    const orc_problem *pb = (const orc_problem *)ctx;
    double c = 0.0, gg[6] = {0, 0, 0, 0, 0, 0}, HH[36]; memset(HH, 0, sizeof HH);
    for (int i = 0; i < pb->pn; ++i) {
        double r[2], J[12];
        orc_residual_jet6(&pb->cam, x, pb->pts2d[2 * i], pb->pts2d[2 * i + 1], pb->pts3d[3 * i], pb->pts3d[3 * i + 1], pb->pts3d[3 * i + 2],
                          pb->wgt2d[2 * i], pb->wgt2d[2 * i + 1], r, J);
        c += r[0] * r[0] + r[1] * r[1];
        for (int a = 0; a < 6; ++a) { gg[a] += J[a] * r[0] + J[6 + a] * r[1];
            for (int b = 0; b < 6; ++b) HH[6 * a + b] += J[a] * J[b] + J[6 + a] * J[6 + b]; }
    }
    *cost = 0.5 * c;
    int ok = isfinite(*cost);
    for (int a = 0; a < 6; ++a) { g[a] = gg[a]; ok = ok && isfinite(gg[a]); }
    for (int a = 0; a < 36; ++a) { H[a] = HH[a]; ok = ok && isfinite(HH[a]); }
    return ok;
}
static int orc_spd_inverse_n(int n, const double *H, double *inv) {
    for (int c = 0; c < n; ++c) {
        double e[ORC_MAXN] = {0}, x[ORC_MAXN]; e[c] = 1.0;
        if (!orc_chol_solve(n, H, e, x)) return 0;
        for (int r = 0; r < n; ++r) inv[n * r + c] = x[r];
    }
    for (int i = 0; i < n * n; ++i) if (!isfinite(inv[i])) return 0;
    return 1;
}
/* one object, host fp64 buffers (the 6-DoF analogue of ext.h's pnp_uncert); diag: iters, why, termination, init/final cost */
void orc_pnp6_uncert(double *pts2d, double *pts3d, double *wgt2d, double *K, double *init_pose6, int *result_val, double *result_pose6,
                     double *result_cov36, double *result_tr, int pn, double *clips, double *diag) {
    orc_problem pb;
    pb.cam.fx = K[0]; pb.cam.fy = K[4]; pb.cam.cx = K[2]; pb.cam.cy = K[5];
    pb.cam.z_min = clips[0]; pb.cam.u_min = clips[1]; pb.cam.u_max = clips[2]; pb.cam.v_min = clips[3]; pb.cam.v_max = clips[4];
    pb.pn = pn; pb.pts2d = pts2d; pb.pts3d = pts3d; pb.wgt2d = wgt2d;
    orc_lm_summary sm;
    orc_lm_n(6, orc_eval6_cb, &pb, init_pose6, result_pose6, &sm);
    *result_val = (sm.termination == ORC_CONVERGENCE || sm.termination == ORC_NO_CONVERGENCE) ? 1 : 0;
    *result_tr = sm.radius;
    if (diag) { diag[0] = sm.num_iterations; diag[1] = sm.why; diag[2] = sm.termination; diag[3] = sm.initial_cost; diag[4] = sm.final_cost; }
    if (*result_val && result_cov36) {
        double cost, g[6], H[36], inv[36];
        int ok = orc_eval6_cb(&pb, result_pose6, &cost, g, H) && orc_spd_inverse_n(6, H, inv);
        *result_val = ok ? 1 : 0;
        if (ok) memcpy(result_cov36, inv, sizeof inv);
    }
}
int orc_eval6(double *pts2d, double *pts3d, double *wgt2d, double *K, double *pose6, int pn, double *clips, double *cost, double *g, double *H,
              double *res /* nullable pn*2 */, double *jac /* nullable pn*12 */) {
    orc_problem pb;
    pb.cam.fx = K[0]; pb.cam.fy = K[4]; pb.cam.cx = K[2]; pb.cam.cy = K[5];
    pb.cam.z_min = clips[0]; pb.cam.u_min = clips[1]; pb.cam.u_max = clips[2]; pb.cam.v_min = clips[3]; pb.cam.v_max = clips[4];
    pb.pn = pn; pb.pts2d = pts2d; pb.pts3d = pts3d; pb.wgt2d = wgt2d;
    if (res && jac) for (int i = 0; i < pn; ++i)
        orc_residual_jet6(&pb.cam, pose6, pts2d[2 * i], pts2d[2 * i + 1], pts3d[3 * i], pts3d[3 * i + 1], pts3d[3 * i + 2], wgt2d[2 * i], wgt2d[2 * i + 1], res + 2 * i, jac + 12 * i);
    return orc_eval6_cb(&pb, pose6, cost, g, H);
}
/* batch refinement as the product's two-launch path does it: for each object the 6-DoF LM starts from the 4-DoF result
 * (r = (0, yaw, 0), t) on the points of the final inlier mask; outputs float32 pose6 (B,6), cov6 (B,36), valid (B). */
void orc_pnp6_refine_batch(const float *x2d, const float *istd, const float *x3d, const float *K, int Kb, const float *u_range, const float *v_range, int Rb,
                           const uint8_t *mask, const float *pose4, const uint8_t *valid4, int B, int P, double z_min, int num_threads,
                           uint8_t *valid, float *pose6, float *cov6, float *diag /* nullable B,2: iters, why */) {
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#pragma omp parallel for schedule(dynamic, 4) if (num_threads != 1)
#endif
    for (int b = 0; b < B; ++b) {
        double *b2 = (double *)malloc(sizeof(double) * (size_t)P * 7), *b3 = b2 + 2 * (size_t)P, *bw = b3 + 3 * (size_t)P;
        int m = 0;
        for (int p = 0; p < P; ++p) if (mask[(size_t)b * P + p]) { const size_t q = (size_t)b * P + p;
            b2[2 * m] = x2d[2 * q]; b2[2 * m + 1] = x2d[2 * q + 1]; b3[3 * m] = x3d[3 * q]; b3[3 * m + 1] = x3d[3 * q + 1]; b3[3 * m + 2] = x3d[3 * q + 2];
            bw[2 * m] = istd[2 * q]; bw[2 * m + 1] = istd[2 * q + 1]; ++m; }
        const float *Kf = K + (Kb == 1 ? 0 : (size_t)b * 9), *ur = u_range + (Rb == 1 ? 0 : (size_t)b * 2), *vr = v_range + (Rb == 1 ? 0 : (size_t)b * 2);
        double Kd[9]; for (int i = 0; i < 9; ++i) Kd[i] = Kf[i];
        double clips[5] = { z_min, ur[0], ur[1], vr[0], vr[1] };
        double init[6] = { 0.0, pose4[4 * b], 0.0, pose4[4 * b + 1], pose4[4 * b + 2], pose4[4 * b + 3] }, out[6], cov[36], tr, dg[5] = {0, 0, 0, 0, 0};
        int val = 0;
        for (int i = 0; i < 36; ++i) cov[i] = (i % 7 == 0) ? 1.0 : 0.0;
        memcpy(out, init, sizeof out);
        if (valid4[b] && m > 0) orc_pnp6_uncert(b2, b3, bw, Kd, init, &val, out, cov, &tr, m, clips, dg);
        valid[b] = (uint8_t)val;
        for (int i = 0; i < 6; ++i) pose6[6 * b + i] = valid4[b] ? (float)out[i] : 0.0f;
        for (int i = 0; i < 36; ++i) cov6[36 * (size_t)b + i] = (float)cov[i];
        if (diag) { diag[2 * b] = (float)dg[0]; diag[2 * b + 1] = (float)dg[1]; }
        free(b2);
    }
}

#include "epnp.inc"     /* the reference's own initialiser restated: EPnP inside OpenCV's RANSAC loop */

/* the reference's initialiser on the candidate subset (pnp_uncert_cpu.py:34-58): EPnP/RANSAC when a threshold is given
 * (the mask is narrowed to the RANSAC inliers when more than 4 come back), plain EPnP otherwise; yaw0 = r_vec[1] (:68) */
static int orc_epnp_init(const float *x2d, const float *x3d, uint8_t *mask, int pn, const float *K, int use_ransac, float thr,
                         double init_pose[4], int *count) {
    int n = 0; for (int p = 0; p < pn; ++p) n += mask[p] ? 1 : 0;
    float *o = (float *)calloc(5 * (size_t)(n > 0 ? n : 1), sizeof(float)); float *im = o + 3 * (size_t)n;
    int *idx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)); uint8_t *rm = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int p = 0; p < pn; ++p) if (mask[p]) { idx[m] = p; o[3 * m] = x3d[3 * p]; o[3 * m + 1] = x3d[3 * p + 1]; o[3 * m + 2] = x3d[3 * p + 2];
                                                im[2 * m] = x2d[2 * p]; im[2 * m + 1] = x2d[2 * p + 1]; ++m; }
    double rvec[3], tvec[3], R[9]; int ok;
    if (use_ransac) {
        ok = epnp_solve_ransac(o, im, n, K, thr, 30, rvec, tvec, rm, NULL);
        int ninl = 0; if (ok) for (int i = 0; i < n; ++i) ninl += rm[i];
        if (ok && ninl > 4) { for (int i = 0; i < n; ++i) mask[idx[i]] = rm[i]; n = ninl; }
    } else if (n >= 4) {
        epnp_solve(o, im, n, K[0], K[4], K[2], K[5], R, tvec, 0, 1); epnp_rodrigues_to_vec(R, rvec); ok = 1;
    } else ok = 0;
    if (ok) { init_pose[0] = rvec[1]; init_pose[1] = tvec[0]; init_pose[2] = tvec[1]; init_pose[3] = tvec[2];
              ok = isfinite(init_pose[0]) && isfinite(init_pose[1]) && isfinite(init_pose[2]) && isfinite(init_pose[3]); }
    if (count) *count = n;
    free(o); free(idx); free(rm);
    return ok;
}

/* ------------------------------------------------------------------------------------------
 * R5 + R6 for a batch: per object  mask0 -> (count>4 ? subset : all) -> K0 (or given init) ->
 * LM on inliers (inlier_opt_only) -> float32 pose -> torch-semantics J^T J on ALL points masked by
 * the final inlier mask -> inverse.  (pnp_uncert_cpu.py:11-125, pnp_uncert.py:45-85.)
 * Inputs contiguous float32 (B,P,2),(B,P,2),(B,P,3); K (Kb,9), ranges (Rb,2) broadcast when Kb/Rb==1.
 * mask: in = istd inlier mask from the host (numpy) stage, out = final inlier mask.
 * ---------------------------------------------------------------------------------------- */
static int orc_cov_waves = 4;   /* waves per object whose summation tree the covariance Hessian follows (orc_cov_hessian_spec); 0 = the sequential order of rounds 1-5 */
void orc_set_cov_waves(int w) { orc_cov_waves = w < 0 ? 0 : (w > 8 ? 8 : w); }

static void orc_one_object(const float *x2d, const float *istd, const float *x3d, const float *K,
                           const float *ur, const float *vr, const float *thr, const double *init,
                           int pn, double z_min, int inlier_opt_only, int n_hyp, int init_mode /* 0 = K0, 1 = EPnP/RANSAC restatement */,
                           uint8_t *mask, uint8_t *valid, float *pose, float *cov, float *tr, float *diag, double *init_out /* nullable 4 */,
                           double *pose64 /* nullable 4: the LM's fp64 iterate */) {
    int cnt = 0; for (int p = 0; p < pn; ++p) cnt += mask[p] ? 1 : 0;
    if (!(cnt > 4)) { for (int p = 0; p < pn; ++p) mask[p] = 1; }                       /* pnp_uncert_cpu.py:23-32 */
    double init_pose[4] = {0, 0, 0, 0}; int ok, bh = -1, bc = 0;
    if (init) { memcpy(init_pose, init, sizeof init_pose); ok = 1; for (int p = 0; p < pn; ++p) bc += mask[p] ? 1 : 0; }
    else if (init_mode == 1) ok = orc_epnp_init(x2d, x3d, mask, pn, K, thr != NULL, thr ? *thr : 0.0f, init_pose, &bc);
    else ok = orc_k0_init(x2d, x3d, mask, pn, K, thr != NULL, thr ? *thr : 0.0f, n_hyp, init_pose, &bh, &bc);
    if (init_out) for (int j = 0; j < 4; ++j) init_out[j] = ok ? init_pose[j] : 0.0;
    double res_pose[4] = {0, 0, 0, 0}, res_tr = 0.0; int res_val = 0; double dg[6] = {0, 0, 0, 0, 0, 0};
    if (ok) {
        /* gather what LM sees, cast to float64 (pnp_uncert_cpu.py:62-81) */
        double *b2 = (double *)malloc(sizeof(double) * (size_t)pn * 7); double *b3 = b2 + 2 * (size_t)pn, *bw = b3 + 3 * (size_t)pn;
        int m = 0;
        for (int p = 0; p < pn; ++p) if (!inlier_opt_only || mask[p]) {
            b2[2 * m] = x2d[2 * p]; b2[2 * m + 1] = x2d[2 * p + 1];
            b3[3 * m] = x3d[3 * p]; b3[3 * m + 1] = x3d[3 * p + 1]; b3[3 * m + 2] = x3d[3 * p + 2];
            bw[2 * m] = istd[2 * p]; bw[2 * m + 1] = istd[2 * p + 1]; ++m; }
        double Kd[9]; for (int i = 0; i < 9; ++i) Kd[i] = K[i];
        double clips[5] = { z_min, ur[0], ur[1], vr[0], vr[1] };
        orc_pnp_uncert_diag(b2, b3, bw, Kd, init_pose, &res_val, res_pose, NULL, &res_tr, m, clips, dg);
        free(b2);
    }
    /* float32 outputs (pnp_uncert_cpu.py:108-125) */
    for (int j = 0; j < 4; ++j) pose[j] = ok ? (float)res_pose[j] : 0.0f;
    if (pose64) for (int j = 0; j < 4; ++j) pose64[j] = ok ? res_pose[j] : 0.0;
    *tr = ok ? (float)res_tr : 0.0f;
    *valid = (uint8_t)(ok && res_val);
    if (diag) { diag[0] = (float)dg[0]; diag[1] = (float)dg[4]; diag[2] = ok ? (float)dg[1] : 8.0f /* initialiser failed */; diag[3] = (float)bc; }
    /* covariance at the float32 pose, all points, final mask (pnp_uncert.py:71-85) */
    {
        double *d2 = (double *)malloc(sizeof(double) * (size_t)pn * 7); double *d3 = d2 + 2 * (size_t)pn, *dw = d3 + 3 * (size_t)pn;
        for (int i = 0; i < 2 * pn; ++i) { d2[i] = x2d[i]; dw[i] = istd[i]; }
        for (int i = 0; i < 3 * pn; ++i) d3[i] = x3d[i];
        double Kd[9]; for (int i = 0; i < 9; ++i) Kd[i] = K[i];
        double urd[2] = { ur[0], ur[1] }, vrd[2] = { vr[0], vr[1] }, td[3] = { pose[1], pose[2], pose[3] }, H[16], C[16];
        if (orc_cov_waves > 0) orc_cov_hessian_spec(Kd, z_min, urd, vrd, (double)pose[0], td, d3, dw, mask, pn, orc_cov_waves, H);      /* the kernel's specified order */
        else orc_torch_jacobian(Kd, z_min, urd, vrd, (double)pose[0], td, d2, d3, dw, mask, pn, NULL, NULL, H);                            /* sequential (rounds 1-5) */
        if (!orc_pose_cov(H, C)) *valid = 0;
        for (int i = 0; i < 16; ++i) cov[i] = (float)C[i];
        free(d2);
    }
}

void orc_u2d_pnp_batch_ex(const float *x2d, const float *istd, const float *x3d,
                          const float *K, int Kb, const float *u_range, const float *v_range, int Rb,
                          const float *ransac_thr /*nullable (B)*/, const double *init_pose /*nullable (B,4)*/,
                          int B, int P, double z_min, int inlier_opt_only, int n_hyp, int init_mode, int num_threads,
                          uint8_t *mask /*B,P in/out*/, uint8_t *valid /*B*/, float *pose /*B,4*/,
                          float *cov /*B,16*/, float *tr /*B*/, float *diag /*nullable B,4*/, double *init_out /*nullable B,4*/,
                          double *pose64 /*nullable B,4*/) {
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#pragma omp parallel for schedule(dynamic, 4) if (num_threads != 1)
#endif
    for (int b = 0; b < B; ++b) {
        orc_one_object(x2d + (size_t)b * P * 2, istd + (size_t)b * P * 2, x3d + (size_t)b * P * 3,
                       K + (Kb == 1 ? 0 : (size_t)b * 9), u_range + (Rb == 1 ? 0 : (size_t)b * 2), v_range + (Rb == 1 ? 0 : (size_t)b * 2),
                       ransac_thr ? ransac_thr + b : NULL, init_pose ? init_pose + (size_t)b * 4 : NULL,
                       P, z_min, inlier_opt_only, n_hyp, init_mode,
                       mask + (size_t)b * P, valid + b, pose + (size_t)b * 4, cov + (size_t)b * 16, tr + b, diag ? diag + (size_t)b * 4 : NULL,
                       init_out ? init_out + (size_t)b * 4 : NULL, pose64 ? pose64 + (size_t)b * 4 : NULL);
    }
}

void orc_u2d_pnp_batch(const float *x2d, const float *istd, const float *x3d,
                       const float *K, int Kb, const float *u_range, const float *v_range, int Rb,
                       const float *ransac_thr /*nullable (B)*/, const double *init_pose /*nullable (B,4)*/,
                       int B, int P, double z_min, int inlier_opt_only, int n_hyp, int num_threads,
                       uint8_t *mask /*B,P in/out*/, uint8_t *valid /*B*/, float *pose /*B,4*/,
                       float *cov /*B,16*/, float *tr /*B*/, float *diag /*nullable B,4*/) {
    orc_u2d_pnp_batch_ex(x2d, istd, x3d, K, Kb, u_range, v_range, Rb, ransac_thr, init_pose, B, P, z_min, inlier_opt_only, n_hyp, 0,
                         num_threads, mask, valid, pose, cov, tr, diag, NULL, NULL);
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

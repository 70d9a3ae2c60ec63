// monorun_pnp.hip — gfx950 (MI355X / CDNA4) kernels + C ABI for MonoRUn's uncertainty-aware PnP.
//
// One workgroup of WPO wavefronts (64 lanes each) owns one object:
//   stage 0  coalesced load of the object's (x2d, istd, X3d) tile into an SoA image in LDS
//   stage 1  istd inlier mask with numpy's float32 summation order          (pnp_uncert_cpu.py:164-168)
//   stage 2  K0: deterministic consensus initialiser (replaces cv2 EPnP/RANSAC, pnp_uncert_cpu.py:35-58)
//   stage 3  trust-region Levenberg–Marquardt, Ceres-1.14 semantics, fp64    (pnp_uncert_cpu.cpp:24-51,245-292)
//   stage 4  pose covariance inverse(J^T J), torch masking semantics        (jacobian.py:48-98, hessian.py:67-87,
//                                                                            pnp_uncert.py:71-85)
// The tile stays in LDS across every LM iteration; per iteration each lane accumulates the 14 non-zero
// scalars of {J^T J, J^T r, cost} for its points, the wave reduces them by shuffles, and every lane
// redundantly solves the damped 4x4 system (uniform control flow, no divergence inside a workgroup).
// No MFMA: J has 50 % structural zeros and the contraction is 4x(2P)x4 — the cost is generating J.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "monorun_pnp.h"

namespace {

constexpr int kMaxLeaves = 64;      // numpy pairwise-sum leaves (blocks of <=128) supported per object
constexpr int kHyp = 32;            // K0 hypotheses (the reference's RANSAC runs 30 iterations)
constexpr uint32_t kK0Seed = 0x9E3779B9u;
constexpr int kRedN = 24;           // doubles per wave in the cross-wave reduction scratch

// numpy's pairwise summation tree for a length-P contiguous float32 reduction, built on the host.
struct PairwisePlan {
    int n_leaves, n_prog;
    uint16_t leaf_off[kMaxLeaves];
    uint16_t leaf_len[kMaxLeaves];
    int8_t prog[2 * kMaxLeaves];    // postfix program: >=0 push leaf sum, -1 add the two top entries
};

struct PnpArgs {
    const void *x2d, *istd, *x3d;
    long long s2[3], sw[3], s3[3];            // element strides (b, p, c)
    const void *K; int K_stride; int K_f64;    // K_stride 0 (broadcast) or 9
    const void *ur, *vr; int r_stride; int r_f64;
    const float *ransac_thr;
    const double *init_pose;
    int B, P, Ppad;
    double z_min; float istd_thres; int inlier_opt_only; int flags; int mean_mode;
    uint8_t *valid; float *pose; float *cov; float *tr; uint8_t *mask; float *diag;
    double *pose64, *cov64, *tr64;            // legacy per-object ABI outputs (nullable)
    unsigned long long *stamps;               // debug: (B,10) s_memtime stamps + HW_ID + XCC_ID per stage (nullable)
    PairwisePlan plan;
};

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
__device__ __forceinline__ double to_f(double v) { return v; }

template <typename T> struct Store { using type = float; };
template <> struct Store<double> { using type = double; };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Sum N doubles over the whole workgroup; every thread gets bit-identical totals.
template <int WPO, int N>
__device__ __forceinline__ void block_sum(double (&a)[N], double *red /* [WPO][kRedN] */) {
#pragma unroll
    for (int k = 0; k < N; ++k) a[k] = wave_sum(a[k]);
    if (WPO > 1) {
        const int w = threadIdx.x >> 6;
        __syncthreads();                              // previous readers of `red` are done
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < N; ++k) red[w * kRedN + k] = a[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double s = red[k];
#pragma unroll
            for (int ww = 1; ww < WPO; ++ww) s += red[ww * kRedN + k];
            a[k] = s;
        }
    }
}

// Cholesky solve of an n x n SPD system (row-major full storage), fixed operation order with explicit
// fma so that the K0 hypotheses are reproducible against the CPU oracle.  false on a non-positive pivot.
template <int n>
__device__ __forceinline__ bool chol_solve(const double (&A)[n * n], const double (&b)[n], double (&x)[n]) {
#pragma clang fp contract(off)
    double L[n * n];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = A[i * n + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s = fma(-L[i * n + k], L[j * n + k], s);
            if (i == j) {
                if (!(s > 0.0) || !isfinite(s)) ok = false;
                L[i * n + i] = sqrt(s);
            } else {
                L[i * n + j] = s / L[j * n + j];
            }
        }
    }
    double y[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s = fma(-L[i * n + k], y[k], s);
        y[i] = s / L[i * n + i];
    }
#pragma unroll
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < n; ++k) s = fma(-L[k * n + i], x[k], s);
        x[i] = s / L[i * n + i];
    }
    return ok;
}

__device__ __forceinline__ bool spd_inverse4(const double (&H)[16], double (&inv)[16]) {
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double e[4] = {0.0, 0.0, 0.0, 0.0}, x[4];
        e[c] = 1.0;
        ok = chol_solve<4>(H, e, x) && ok;
#pragma unroll
        for (int r = 0; r < 4; ++r) inv[4 * r + c] = x[r];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) ok = ok && isfinite(inv[i]);
    return ok;
}

// ------------------------------------------------------------------------------------ K0 pieces --
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

struct K4 { double fx, fy, cx, cy; };

// one correspondence into the linear 4-DoF system in theta = (cos, sin, tx, ty, tz):
//   (u-cx)(-s x + c z + tz) = fx (c x + s z + tx),  (v-cy)(-s x + c z + tz) = fy (y + ty)
__device__ __forceinline__ void lin5_add(const K4 &k, float u, float v, float x, float y, float z,
                                         double (&n5)[15], double (&m5)[5]) {
#pragma clang fp contract(off)
    const double a = (double)u - k.cx, b = (double)v - k.cy;
    const double X = x, Y = y, Z = z;
    const double ru[5] = { fma(a, Z, -(k.fx * X)), fma(-a, X, -(k.fx * Z)), -k.fx, 0.0, a };
    const double rv[5] = { b * Z, -(b * X), 0.0, -k.fy, b };
    const double rhs_v = k.fy * Y;
    int q = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int j = i; j < 5; ++j, ++q) n5[q] = fma(rv[i], rv[j], fma(ru[i], ru[j], n5[q]));
        m5[i] = fma(rv[i], rhs_v, m5[i]);
    }
}
__device__ __forceinline__ void lin3_add(const K4 &k, double c, double s, float u, float v, float x, float y, float z,
                                         double (&n3)[4], double (&m3)[3]) {
#pragma clang fp contract(off)
    const double a = (double)u - k.cx, b = (double)v - k.cy;
    const double X = x, Y = y, Z = z;
    const double Xr = fma(c, X, s * Z), Zr = fma(c, Z, -(s * X));
    const double bu = fma(k.fx, Xr, -(a * Zr));
    const double bv = fma(k.fy, Y, -(b * Zr));
    n3[0] += 1.0; n3[1] += a; n3[2] += b; n3[3] = fma(b, b, fma(a, a, n3[3]));
    m3[0] = fma(-k.fx, bu, m3[0]); m3[1] = fma(-k.fy, bv, m3[1]); m3[2] = fma(b, bv, fma(a, bu, m3[2]));
}
__device__ __forceinline__ bool lin5_solve(const double (&n5)[15], const double (&m5)[5], double &c, double &s) {
#pragma clang fp contract(off)
    double A[25], th[5];
    int q = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = i; j < 5; ++j, ++q) { A[5 * i + j] = n5[q]; A[5 * j + i] = n5[q]; }
    bool ok = chol_solve<5>(A, m5, th);
    const double rho2 = fma(th[0], th[0], th[1] * th[1]);
    ok = ok && (rho2 > 1e-12) && isfinite(rho2);
    const double inv = 1.0 / sqrt(rho2);
    c = th[0] * inv; s = th[1] * inv;
    return ok;
}
__device__ __forceinline__ bool lin3_solve(const K4 &k, const double (&n3)[4], const double (&m3)[3], double (&t)[3]) {
#pragma clang fp contract(off)
    const double A[9] = { k.fx * k.fx * n3[0], 0.0, -k.fx * n3[1],
                          0.0, k.fy * k.fy * n3[0], -k.fy * n3[2],
                          -k.fx * n3[1], -k.fy * n3[2], n3[3] };
    bool ok = chol_solve<3>(A, m3, t);
    return ok && isfinite(t[0]) && isfinite(t[1]) && isfinite(t[2]);
}
__device__ __forceinline__ bool consensus(float c, float s, float tx, float ty, float tz, float fx, float fy,
                                          float a, float b, float x, float y, float z, float thr) {
#pragma clang fp contract(off)
    const float Xc = fmaf(c, x, fmaf(s, z, tx));
    const float Zc = fmaf(c, z, fmaf(-s, x, tz));
    const float Yc = y + ty;
    const float eu = fmaf(-a, Zc, fx * Xc);
    const float ev = fmaf(-b, Zc, fy * Yc);
    const float e2 = fmaf(eu, eu, ev * ev);
    const float lim = thr * Zc;
    return (Zc > 0.0f) && (e2 <= lim * lim);
}

// ------------------------------------------------------------------------------------ LM pieces --
struct Cam { double fx, fy, cx, cy, zmin, umin, umax, vmin, vmax; };

// accumulator slots: 0..8 = H00 H01 H02 H03 H11 H13 H22 H23 H33 (H12 == 0 structurally), 9..12 = g, 13 = sum r^2
constexpr int kAcc = 14;

// Residual + Jacobian of one correspondence with the semantics of Ceres autodiff on
// ReprojectionErrorArray (pnp_uncert_cpu.cpp:24-51): a clamped quantity becomes a constant, so the
// z-clamp removes only dZ, and a u/v clamp removes that whole row.
__device__ __forceinline__ void eval_point(const Cam &k, double c, double s, double tx, double ty, double tz,
                                           double u, double v, double wu, double wv, double x, double y, double z,
                                           double (&acc)[kAcc]) {
    const double Bv = c * x + s * z;               // Xc - tx ;  dZc/dyaw = -Bv
    const double A = c * z - s * x;                // Zc - tz ;  dXc/dyaw =  A
    const double Xc = Bv + tx, Yc = y + ty, Zc = A + tz;
    const bool zc = Zc < k.zmin;
    const double Zu = zc ? k.zmin : Zc;
    const double iz = 1.0 / Zu;
    const double px = k.fx * Xc * iz, py = k.fy * Yc * iz;
    double pu = px + k.cx, pv = py + k.cy;
    const bool ucl = (pu < k.umin) || (pu > k.umax);
    const bool vcl = (pv < k.vmin) || (pv > k.vmax);
    pu = (pu < k.umin) ? k.umin : ((pu > k.umax) ? k.umax : pu);
    pv = (pv < k.vmin) ? k.vmin : ((pv > k.vmax) ? k.vmax : pv);
    const double ru = wu * (pu - u), rv = wv * (pv - v);
    const double dZy = zc ? 0.0 : -Bv, dZt = zc ? 0.0 : 1.0;
    const double wuz = ucl ? 0.0 : wu * iz, wvz = vcl ? 0.0 : wv * iz;
    const double ju0 = wuz * (k.fx * A - px * dZy);
    const double ju1 = wuz * k.fx;
    const double ju3 = -wuz * px * dZt;
    const double jv0 = -wvz * py * dZy;
    const double jv2 = wvz * k.fy;
    const double jv3 = -wvz * py * dZt;
    acc[0] += ju0 * ju0 + jv0 * jv0;
    acc[1] += ju0 * ju1;
    acc[2] += jv0 * jv2;
    acc[3] += ju0 * ju3 + jv0 * jv3;
    acc[4] += ju1 * ju1;
    acc[5] += ju1 * ju3;
    acc[6] += jv2 * jv2;
    acc[7] += jv2 * jv3;
    acc[8] += ju3 * ju3 + jv3 * jv3;
    acc[9] += ju0 * ru + jv0 * rv;
    acc[10] += ju1 * ru;
    acc[11] += jv2 * rv;
    acc[12] += ju3 * ru + jv3 * rv;
    acc[13] += ru * ru + rv * rv;
}

struct Eval { double cost; double g[4]; double H[16]; bool ok; };

template <int WPO, typename S>
__device__ __forceinline__ void evaluate(const Cam &k, const double (&x)[4], int P, int Ppad, bool use_mask,
                                         const S *su, const S *sv, const S *swu, const S *swv,
                                         const S *sx, const S *sy, const S *sz, const uint8_t *smask,
                                         double *red, Eval &e) {
    constexpr int NT = 64 * WPO;
    double sn, cs;
    sincos(x[0], &sn, &cs);
    double acc[kAcc];
#pragma unroll
    for (int i = 0; i < kAcc; ++i) acc[i] = 0.0;
#pragma unroll 2
    for (int p = threadIdx.x; p < Ppad; p += NT) {
        // points outside the inlier set are not part of the problem at all (pnp_uncert_cpu.py:62-66):
        // skip them instead of zero-weighting them so that a NaN correspondence there cannot poison the sums
        if (use_mask ? (smask[p] != 0) : (p < P))
            eval_point(k, cs, sn, x[1], x[2], x[3], (double)su[p], (double)sv[p], (double)swu[p], (double)swv[p],
                       (double)sx[p], (double)sy[p], (double)sz[p], acc);
    }
    block_sum<WPO, kAcc>(acc, red);
    e.cost = 0.5 * acc[13];
    e.g[0] = acc[9]; e.g[1] = acc[10]; e.g[2] = acc[11]; e.g[3] = acc[12];
    e.H[0] = acc[0]; e.H[1] = acc[1]; e.H[2] = acc[2]; e.H[3] = acc[3];
    e.H[4] = acc[1]; e.H[5] = acc[4]; e.H[6] = 0.0;    e.H[7] = acc[5];
    e.H[8] = acc[2]; e.H[9] = 0.0;    e.H[10] = acc[6]; e.H[11] = acc[7];
    e.H[12] = acc[3]; e.H[13] = acc[5]; e.H[14] = acc[7]; e.H[15] = acc[8];
    bool ok = isfinite(e.cost);
#pragma unroll
    for (int i = 0; i < 13; ++i) ok = ok && isfinite(acc[i]);
    e.ok = ok;
}

enum { WHY_GRADIENT = 1, WHY_PARAMETER = 2, WHY_FUNCTION = 3, WHY_MAXITER = 4, WHY_MINRADIUS = 5,
       WHY_INVALID = 6, WHY_EVALFAIL = 7, WHY_K0FAIL = 8 };

struct LmResult { double x[4]; double radius; double cost; int iters; int why; bool usable; };

// Ceres 1.14 TrustRegionMinimizer + LevenbergMarquardtStrategy with default options (only
// linear_solver_type = DENSE_QR is set by the reference, pnp_uncert_cpu.cpp:270-271); the QR solve of
// [J; D] y = [r; 0] is done through its (Jacobi-scaled, 4x4, fp64) normal equations.
template <int WPO, typename S>
__device__ void lm_solve(const Cam &k, const double (&init)[4], int P, int Ppad, bool use_mask,
                         const S *su, const S *sv, const S *swu, const S *swv,
                         const S *sx, const S *sy, const S *sz, const uint8_t *smask,
                         double *red, LmResult &r) {
    const int max_num_iterations = 50;
    const double max_radius = 1e16, min_radius = 1e-32;
    const double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;

    double x[4] = { init[0], init[1], init[2], init[3] };
#pragma unroll
    for (int j = 0; j < 4; ++j) r.x[j] = x[j];
    double radius = 1e4, decrease_factor = 2.0;
    r.iters = 0; r.why = 0; r.usable = false; r.radius = radius; r.cost = 0.0;

    Eval cur;
    evaluate<WPO, S>(k, x, P, Ppad, use_mask, su, sv, swu, swv, sx, sy, sz, smask, red, cur);
    if (!cur.ok) { r.why = WHY_EVALFAIL; r.radius = 0.0; return; }
    double scale[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) scale[j] = 1.0 / (1.0 + sqrt(cur.H[5 * j]));
    double x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    bool last_successful = true;
    int iteration = 0, invalid_run = 0;

    for (;;) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (last_successful) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r.x[j] = x[j];
            r.cost = cur.cost;
        }
        r.radius = radius; r.iters = iteration;
        if (iteration >= max_num_iterations) { r.why = WHY_MAXITER; r.usable = true; return; }
        if (last_successful) {
            double gmax = fmax(fmax(fabs(cur.g[0]), fabs(cur.g[1])), fmax(fabs(cur.g[2]), fabs(cur.g[3])));
            if (gmax <= gradient_tolerance) { r.why = WHY_GRADIENT; r.usable = true; return; }
        }
        if (radius <= min_radius) { r.why = WHY_MINRADIUS; r.usable = true; return; }

        ++iteration; last_successful = false;
        // LevenbergMarquardtStrategy::ComputeStep on the column-scaled Jacobian
        double Hs[16], gs[4], A[16], y[4], step[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            gs[a] = cur.g[a] * scale[a];
#pragma unroll
            for (int b = 0; b < 4; ++b) Hs[4 * a + b] = cur.H[4 * a + b] * scale[a] * scale[b];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) A[i] = Hs[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double d = fmin(fmax(Hs[5 * j], min_lm_diagonal), max_lm_diagonal);
            A[5 * j] += d / radius;
        }
        bool step_ok = chol_solve<4>(A, gs, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) { step[j] = -y[j]; step_ok = step_ok && isfinite(step[j]); }
        double model_cost_change = 0.0;
        if (step_ok) {
            double sg = 0.0, sHs = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                sg += step[a] * gs[a];
                double t = 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b) t += Hs[4 * a + b] * step[b];
                sHs += step[a] * t;
            }
            model_cost_change = -(sg + 0.5 * sHs);
            step_ok = model_cost_change > 0.0;
        }
        if (!step_ok) {                                   // HandleInvalidStep
            if (++invalid_run >= 5) { r.why = WHY_INVALID; r.usable = false; r.iters = iteration; return; }
            radius *= 0.5;
            continue;
        }
        invalid_run = 0;
        double cand[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cand[j] = x[j] + step[j] * scale[j];
        Eval nxt;                                         // cost, and speculatively g/H, at the candidate
        evaluate<WPO, S>(k, cand, P, Ppad, use_mask, su, sv, swu, swv, sx, sy, sz, smask, red, nxt);
        const double cand_cost = nxt.ok ? nxt.cost : 1.7976931348623157e308;
        double step_norm = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double d = x[j] - cand[j]; step_norm += d * d; }
        step_norm = sqrt(step_norm);
        if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) {
            r.why = WHY_PARAMETER; r.usable = true; r.iters = iteration; return; }
        const double cost_change = cur.cost - cand_cost;
        if (fabs(cost_change) <= function_tolerance * cur.cost) {
            r.why = WHY_FUNCTION; r.usable = true; r.iters = iteration; return; }
        const double relative_decrease = cost_change / model_cost_change;
        if (relative_decrease > min_relative_decrease) {  // HandleSuccessfulStep + StepAccepted
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = cand[j];
            x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
            cur = nxt;
            last_successful = true;
            const double t = 2.0 * relative_decrease - 1.0;
            radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            radius = fmin(max_radius, radius);
            decrease_factor = 2.0;
        } else {                                          // StepRejected
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <typename T, typename S>
__device__ __forceinline__ void load_tile(const T *src, long long sb, long long sp, long long sc, int C, int b, int P,
                                          int Ppad, int NT, S *dst /* [C][Ppad] */, S pad_last) {
    const T *base = src + (long long)b * sb;
    if (sc == 1 && sp == C) {
        // contiguous (P, C) block: flat coalesced sweep, de-interleaved into the SoA image
        const int n = P * C;
        for (int e = threadIdx.x; e < n; e += NT) {
            const int p = e / C, c = e - p * C;
            dst[c * Ppad + p] = (S)to_f(base[e]);
        }
    } else {
        // channel-planar (sp == 1: lane-consecutive points are address-consecutive) or any other strides
        for (int c = 0; c < C; ++c) {
            const T *bc = base + (long long)c * sc;
            for (int p = threadIdx.x; p < P; p += NT) dst[c * Ppad + p] = (S)to_f(bc[(long long)p * sp]);
        }
    }
    for (int c = 0; c < C; ++c)
        for (int p = P + threadIdx.x; p < Ppad; p += NT) dst[c * Ppad + p] = (c == C - 1) ? pad_last : (S)0;
}

__device__ __forceinline__ double ld_k(const void *p, int f64, long long i) {
    return f64 ? ((const double *)p)[i] : (double)((const float *)p)[i];
}

template <typename T, int WPO>
__global__ void __launch_bounds__(64 * WPO) pnp_uncert_kernel(const PnpArgs a) {
    using S = typename Store<T>::type;
    constexpr int NT = 64 * WPO;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int P = a.P, Ppad = a.Ppad;

    extern __shared__ __align__(16) unsigned char smem[];
#define MR_STAMP(i) do { if (a.stamps && tid == 0) a.stamps[(long long)b * 10 + (i)] = __builtin_readcyclecounter(); } while (0)
    MR_STAMP(0);
    if (a.stamps && tid == 0) { a.stamps[(long long)b * 10 + 8] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); a.stamps[(long long)b * 10 + 9] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }
    double *red = (double *)smem;                                   // [WPO][kRedN]
    unsigned long long *sball = (unsigned long long *)(red + WPO * kRedN);   // [Ppad/64]
    S *su = (S *)(sball + ((Ppad / 64 + 1) & ~1));                  // keep the tile 16-byte aligned
    S *sv = su + Ppad, *swu = sv + Ppad, *swv = swu + Ppad, *sx = swv + Ppad, *sy = sx + Ppad, *sz = sy + Ppad;
    float *shyp = (float *)(sz + Ppad);                             // [kHyp][8]
    int *scnt = (int *)(shyp + kHyp * 8);                           // [WPO][kHyp]
    float *spw = (float *)(scnt + WPO * kHyp);                      // [2][kMaxLeaves*8] pairwise partials
    float *sleaf = spw + 2 * kMaxLeaves * 8;                        // [2][kMaxLeaves]
    float *sstk = sleaf + 2 * kMaxLeaves;                           // [2][16] + [2] results
    uint16_t *slist = (uint16_t *)(sstk + 2 * 16 + 2);              // [Ppad]
    uint8_t *smask = (uint8_t *)(slist + Ppad);                     // [Ppad]

    // ---------------------------------------------------------------- stage 0: tile -> LDS (SoA)
    load_tile<T, S>((const T *)a.x2d, a.s2[0], a.s2[1], a.s2[2], 2, b, P, Ppad, NT, su, (S)0);
    load_tile<T, S>((const T *)a.istd, a.sw[0], a.sw[1], a.sw[2], 2, b, P, Ppad, NT, swu, (S)0);
    load_tile<T, S>((const T *)a.x3d, a.s3[0], a.s3[1], a.s3[2], 3, b, P, Ppad, NT, sx, (S)1);   // padded points: z = 1, w = 0
    const long long ko = (long long)b * a.K_stride, ro = (long long)b * a.r_stride;
    double Kd[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Kd[i] = ld_k(a.K, a.K_f64, ko + i);
    Cam cam;
    cam.fx = Kd[0]; cam.fy = Kd[4]; cam.cx = Kd[2]; cam.cy = Kd[5];      // pnp_uncert_cpu.cpp:265
    cam.zmin = a.z_min;
    cam.umin = ld_k(a.ur, a.r_f64, ro); cam.umax = ld_k(a.ur, a.r_f64, ro + 1);
    cam.vmin = ld_k(a.vr, a.r_f64, ro); cam.vmax = ld_k(a.vr, a.r_f64, ro + 1);
    __syncthreads();

    MR_STAMP(1);
    // ---------------------------------------------------------------- stage 1: istd inlier mask (R4)
    if (a.flags & MR_NO_ISTD_MASK) {
        for (int p = tid; p < Ppad; p += NT) smask[p] = (p < P) ? 1 : 0;
        __syncthreads();
    } else {
        float *sres = sstk + 32;
        if (a.mean_mode == MR_MEAN_SEQUENTIAL) {
            // numpy reduces a C-contiguous (B,P,2) float32 array over axis 1 strictly in p order
            if (tid < 2) {
                const S *w = tid ? swv : swu;
                float s = 0.0f;
                for (int p = 0; p < P; ++p) s += (float)w[p];
                sres[tid] = s;
            }
        } else {
            // numpy pairwise summation (point axis contiguous): 8 strided accumulators per <=128 block
            const int nl = a.plan.n_leaves;
            for (int t = tid; t < 2 * nl * 8; t += NT) {
                const int c = t / (nl * 8), rem = t - c * nl * 8, leaf = rem >> 3, j = rem & 7;
                const S *w = (c ? swv : swu) + a.plan.leaf_off[leaf];
                const int len = a.plan.leaf_len[leaf];
                float acc = 0.0f;
                if (len < 8) {
                    if (j == 0) for (int i = 0; i < len; ++i) acc += (float)w[i];
                } else {
                    acc = (float)w[j];
                    for (int i = 8; i < len - (len & 7); i += 8) acc += (float)w[i + j];
                }
                spw[t] = acc;
            }
            __syncthreads();
            for (int t = tid; t < 2 * nl; t += NT) {
                const int c = t / nl, leaf = t - c * nl;
                const float *r8 = spw + (c * nl + leaf) * 8;
                const int len = a.plan.leaf_len[leaf];
                float res;
                if (len < 8) res = r8[0];
                else {
                    res = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
                    const S *w = (c ? swv : swu) + a.plan.leaf_off[leaf];
                    for (int i = len - (len & 7); i < len; ++i) res += (float)w[i];
                }
                sleaf[c * kMaxLeaves + leaf] = res;
            }
            __syncthreads();
            if (tid < 2) {
                float *stk = sstk + tid * 16;
                int sp = 0;
                for (int i = 0; i < a.plan.n_prog; ++i) {
                    const int op = a.plan.prog[i];
                    if (op >= 0) stk[sp++] = sleaf[tid * kMaxLeaves + op];
                    else { const float r = stk[sp - 2] + stk[sp - 1]; sp -= 2; stk[sp++] = r; }
                }
                sres[tid] = stk[0];
            }
        }
        __syncthreads();
        // mean = sum / P ; threshold = float32(thres) * mean ; both axes must pass (pnp_uncert_cpu.py:164-168)
        const float thr_u = a.istd_thres * (sres[0] / (float)P);
        const float thr_v = a.istd_thres * (sres[1] / (float)P);
        int cnt = 0;
        for (int p = tid; p < Ppad; p += NT) {
            const bool in = (p < P) && ((float)swu[p] >= thr_u) && ((float)swv[p] >= thr_v);
            smask[p] = in ? 1 : 0;
            cnt += in ? 1 : 0;
        }
        double c1[1] = { (double)cnt };
        block_sum<WPO, 1>(c1, red);
        if (!(c1[0] > 4.0)) {                                       // pnp_uncert_cpu.py:23-32
            for (int p = tid; p < Ppad; p += NT) smask[p] = (p < P) ? 1 : 0;
        }
        __syncthreads();
    }

    MR_STAMP(2);
    // ---------------------------------------------------------------- stage 2: K0 initialiser (R5)
    double init[4] = { 0.0, 0.0, 0.0, 0.0 };
    bool init_ok = true;
    float k0_count = 0.0f;
    if (a.init_pose) {
#pragma unroll
        for (int j = 0; j < 4; ++j) init[j] = a.init_pose[(long long)b * 4 + j];
        int cnt = 0;
        for (int p = tid; p < Ppad; p += NT) cnt += smask[p];
        double c1[1] = { (double)cnt };
        block_sum<WPO, 1>(c1, red);
        k0_count = (float)c1[0];
    } else {
        const K4 k4 = { cam.fx, cam.fy, cam.cx, cam.cy };
        const float fxf = (float)cam.fx, fyf = (float)cam.fy, cxf = (float)cam.cx, cyf = (float)cam.cy;
        // ordered compaction of the candidate indices
        for (int p = tid; p < Ppad; p += NT) {
            const unsigned long long bal = __ballot(smask[p] != 0);
            if (lane == 0) sball[p >> 6] = bal;
        }
        __syncthreads();
        int n = 0;
        for (int p = tid; p < Ppad; p += NT) {
            const int chunk = p >> 6;
            int base = 0;
            for (int q = 0; q < chunk; ++q) base += __popcll(sball[q]);
            const unsigned long long bal = sball[chunk];
            if ((bal >> lane) & 1ull) slist[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)p;
        }
        for (int q = 0; q < Ppad / 64; ++q) n += __popcll(sball[q]);
        __syncthreads();
        k0_count = (float)n;
        const bool use_ransac = a.ransac_thr != nullptr;
        if (use_ransac) {
            const float thr = a.ransac_thr[b];
            // hypotheses: lane h of wave 0 solves hypothesis h from 5 stratified samples (fp64, serial)
            if (tid < kHyp) {
                bool ok = n >= 5;
                double c = 0.0, s = 0.0, t[3] = { 0.0, 0.0, 0.0 };
                if (ok) {
                    double n5[15], m5[5], n3[4] = { 0, 0, 0, 0 }, m3[3] = { 0, 0, 0 };
                    int idx[5];
#pragma unroll
                    for (int i = 0; i < 15; ++i) n5[i] = 0.0;
#pragma unroll
                    for (int i = 0; i < 5; ++i) m5[i] = 0.0;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const uint32_t lo = (uint32_t)(((uint64_t)j * (uint64_t)n) / 5u), hi = (uint32_t)(((uint64_t)(j + 1) * (uint64_t)n) / 5u);
                        const uint32_t rr = lo + (uint32_t)(((uint64_t)hash32(kK0Seed + (uint32_t)tid * 8u + (uint32_t)j) * (uint64_t)(hi - lo)) >> 32);
                        idx[j] = slist[rr];
                        const int p = idx[j];
                        lin5_add(k4, (float)su[p], (float)sv[p], (float)sx[p], (float)sy[p], (float)sz[p], n5, m5);
                    }
                    ok = lin5_solve(n5, m5, c, s);
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int p = idx[j];
                        lin3_add(k4, c, s, (float)su[p], (float)sv[p], (float)sx[p], (float)sy[p], (float)sz[p], n3, m3);
                    }
                    ok = lin3_solve(k4, n3, m3, t) && ok;
                }
                const float nanf_ = __int_as_float(0x7fc00000);
                shyp[tid * 8 + 0] = ok ? (float)c : nanf_;
                shyp[tid * 8 + 1] = ok ? (float)s : nanf_;
                shyp[tid * 8 + 2] = ok ? (float)t[0] : nanf_;
                shyp[tid * 8 + 3] = ok ? (float)t[1] : nanf_;
                shyp[tid * 8 + 4] = ok ? (float)t[2] : nanf_;
            }
            __syncthreads();
            MR_STAMP(3);
            // consensus of every hypothesis over the candidates (fp32, fixed operation order)
            int cnt[kHyp];
#pragma unroll
            for (int h = 0; h < kHyp; ++h) cnt[h] = 0;
            for (int p = tid; p < Ppad; p += NT) {
                const float pa = (float)su[p] - cxf, pb = (float)sv[p] - cyf;
                const float px = (float)sx[p], py = (float)sy[p], pz = (float)sz[p];
                const bool m = smask[p] != 0;
#pragma unroll
                for (int h = 0; h < kHyp; ++h) {
                    const float4 hp = *(const float4 *)(shyp + h * 8);
                    const float htz = shyp[h * 8 + 4];
                    const bool in = m && consensus(hp.x, hp.y, hp.z, hp.w, htz, fxf, fyf, pa, pb, px, py, pz, thr);
                    cnt[h] += __popcll(__ballot(in));
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int h = 0; h < kHyp; ++h) scnt[wid * kHyp + h] = cnt[h];
            }
            __syncthreads();
            int best = -1, bestc = 0;
            for (int h = 0; h < kHyp; ++h) {
                int ch = 0;
                for (int w = 0; w < WPO; ++w) ch += scnt[w * kHyp + h];
                if (ch > bestc) { bestc = ch; best = h; }
            }
            k0_count = (float)bestc;
            if (best < 0 || bestc < 5) init_ok = false;
            else {
                const float4 hp = *(const float4 *)(shyp + best * 8);
                const float htz = shyp[best * 8 + 4];
                for (int p = tid; p < Ppad; p += NT) {
                    const bool in = (smask[p] != 0) &&
                        consensus(hp.x, hp.y, hp.z, hp.w, htz, fxf, fyf, (float)su[p] - cxf, (float)sv[p] - cyf,
                                  (float)sx[p], (float)sy[p], (float)sz[p], thr);
                    smask[p] = in ? 1 : 0;
                }
            }
            __syncthreads();
        }
        MR_STAMP(4);
        if (init_ok) {
            // refit on the final set with the same linear solver (fp64 accumulation, tree reduction)
            double acc20[20];
#pragma unroll
            for (int i = 0; i < 20; ++i) acc20[i] = 0.0;
            {
                double n5[15], m5[5];
#pragma unroll
                for (int i = 0; i < 15; ++i) n5[i] = 0.0;
#pragma unroll
                for (int i = 0; i < 5; ++i) m5[i] = 0.0;
                for (int p = tid; p < Ppad; p += NT)
                    if (smask[p]) lin5_add(k4, (float)su[p], (float)sv[p], (float)sx[p], (float)sy[p], (float)sz[p], n5, m5);
#pragma unroll
                for (int i = 0; i < 15; ++i) acc20[i] = n5[i];
#pragma unroll
                for (int i = 0; i < 5; ++i) acc20[15 + i] = m5[i];
            }
            block_sum<WPO, 20>(acc20, red);
            double n5[15], m5[5], c, s;
#pragma unroll
            for (int i = 0; i < 15; ++i) n5[i] = acc20[i];
#pragma unroll
            for (int i = 0; i < 5; ++i) m5[i] = acc20[15 + i];
            init_ok = lin5_solve(n5, m5, c, s);
            double acc7[7] = { 0, 0, 0, 0, 0, 0, 0 };
            {
                double n3[4] = { 0, 0, 0, 0 }, m3[3] = { 0, 0, 0 };
                for (int p = tid; p < Ppad; p += NT)
                    if (smask[p]) lin3_add(k4, c, s, (float)su[p], (float)sv[p], (float)sx[p], (float)sy[p], (float)sz[p], n3, m3);
                acc7[0] = n3[0]; acc7[1] = n3[1]; acc7[2] = n3[2]; acc7[3] = n3[3];
                acc7[4] = m3[0]; acc7[5] = m3[1]; acc7[6] = m3[2];
            }
            block_sum<WPO, 7>(acc7, red);
            const double n3[4] = { acc7[0], acc7[1], acc7[2], acc7[3] }, m3[3] = { acc7[4], acc7[5], acc7[6] };
            double t[3];
            init_ok = lin3_solve(k4, n3, m3, t) && init_ok;
            init[0] = atan2(s, c); init[1] = t[0]; init[2] = t[1]; init[3] = t[2];
        }
    }

    MR_STAMP(5);
    // ---------------------------------------------------------------- stage 3: LM refinement (R1,R3)
    LmResult lm;
    lm.x[0] = lm.x[1] = lm.x[2] = lm.x[3] = 0.0; lm.radius = 0.0; lm.cost = 0.0; lm.iters = 0; lm.why = WHY_K0FAIL; lm.usable = false;
    if (init_ok)
        lm_solve<WPO, S>(cam, init, P, Ppad, a.inlier_opt_only != 0, su, sv, swu, swv, sx, sy, sz, smask, red, lm);
    const bool pose_ok = init_ok;
    float posef[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) posef[j] = pose_ok ? (float)lm.x[j] : 0.0f;       // pnp_uncert_cpu.py:108-125
    bool valid = pose_ok && lm.usable;

    MR_STAMP(6);
    // ---------------------------------------------------------------- stage 4: covariance (R2,R6,R7)
    double cov[16];
    bool have_cov = false;
    if (!(a.flags & MR_COV_NONE)) {
        have_cov = true;
        double H[16];
        if (a.flags & MR_COV_CERES) {
            Eval e;
            double xe[4] = { lm.x[0], lm.x[1], lm.x[2], lm.x[3] };
            evaluate<WPO, S>(cam, xe, P, Ppad, a.inlier_opt_only != 0, su, sv, swu, swv, sx, sy, sz, smask, red, e);
#pragma unroll
            for (int i = 0; i < 16; ++i) H[i] = e.H[i];
        } else {
            // torch semantics at the float32 pose, all points, masked by the final inlier set:
            // zero_mask = z_clip | uv_clip(axis) | outlier ; full upper 2x3 of K (jacobian.py:20-26,52-95)
            const double yaw = (double)posef[0], tx = (double)posef[1], ty = (double)posef[2], tz = (double)posef[3];
            double sn, cs;
            sincos(yaw, &sn, &cs);
            double kr[9], kt[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                kr[3 * r + 0] = Kd[3 * r + 0] * cs - Kd[3 * r + 2] * sn;
                kr[3 * r + 1] = Kd[3 * r + 1];
                kr[3 * r + 2] = Kd[3 * r + 0] * sn + Kd[3 * r + 2] * cs;
                kt[r] = Kd[3 * r + 0] * tx + Kd[3 * r + 1] * ty + Kd[3 * r + 2] * tz;
            }
            const double m1[4] = { Kd[0] * (-sn) + Kd[2] * (-cs), Kd[0] * cs + Kd[2] * (-sn),
                                   Kd[3] * (-sn) + Kd[5] * (-cs), Kd[3] * cs + Kd[5] * (-sn) };
            double hacc[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) hacc[i] = 0.0;
            for (int p = tid; p < Ppad; p += NT) {
                const double X = (double)sx[p], Y = (double)sy[p], Z = (double)sz[p];
                const double un = kr[0] * X + kr[1] * Y + kr[2] * Z + kt[0];
                const double vn = kr[3] * X + kr[4] * Y + kr[5] * Z + kt[1];
                double z = kr[6] * X + kr[7] * Y + kr[8] * Z + kt[2];
                const bool zclip = z < cam.zmin;
                z = zclip ? cam.zmin : z;
                const double iz = 1.0 / z;
                double uv[2] = { un * iz, vn * iz };
                const bool cl[2] = { (uv[0] < cam.umin) || (uv[0] > cam.umax), (uv[1] < cam.vmin) || (uv[1] > cam.vmax) };
                uv[0] = fmax(cam.umin, fmin(cam.umax, uv[0]));
                uv[1] = fmax(cam.vmin, fmin(cam.vmax, uv[1]));
                const bool outl = smask[p] == 0;
                double J[8];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    // jac[zero_mask] = 0 is an assignment (jacobian.py:70,95): select, do not multiply by zero
                    const bool zero = zclip || cl[r] || outl;
                    const double w = (double)(r ? swv[p] : swu[p]) * iz;
                    J[4 * r + 0] = zero ? 0.0 : w * ((m1[2 * r] + uv[r] * cs) * X + (m1[2 * r + 1] + uv[r] * sn) * Z);
                    J[4 * r + 1] = zero ? 0.0 : w * Kd[3 * r + 0];
                    J[4 * r + 2] = zero ? 0.0 : w * Kd[3 * r + 1];
                    J[4 * r + 3] = zero ? 0.0 : w * (Kd[3 * r + 2] - uv[r]);
                }
                int q = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = i; j < 4; ++j, ++q) hacc[q] += J[i] * J[j] + J[4 + i] * J[4 + j];
            }
            block_sum<WPO, 10>(hacc, red);
            int q = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = i; j < 4; ++j, ++q) { H[4 * i + j] = hacc[q]; H[4 * j + i] = hacc[q]; }
        }
        const bool inv_ok = spd_inverse4(H, cov);
        if (!inv_ok) {
            if (a.flags & MR_COV_CERES) have_cov = false;          // result_cov left untouched (pnp_uncert_cpu.cpp:285-290)
            else {
#pragma unroll
                for (int i = 0; i < 16; ++i) cov[i] = (i % 5 == 0) ? 1.0 : 0.0;   // h := I (pnp_uncert.py:83-85)
            }
            valid = false;
        }
    }

    MR_STAMP(7);
    // ---------------------------------------------------------------- outputs
    if (tid == 0) {
        a.valid[b] = valid ? 1 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) a.pose[(long long)b * 4 + j] = posef[j];
        a.tr[b] = pose_ok ? (float)lm.radius : 0.0f;
        if (a.diag) {
            a.diag[(long long)b * 4 + 0] = (float)lm.iters;
            a.diag[(long long)b * 4 + 1] = (float)lm.cost;
            a.diag[(long long)b * 4 + 2] = (float)lm.why;
            a.diag[(long long)b * 4 + 3] = k0_count;
        }
        if (a.pose64) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a.pose64[(long long)b * 4 + j] = lm.x[j];
            a.tr64[b] = lm.radius;
        }
    }
    if (have_cov && tid < 16) {
        a.cov[(long long)b * 16 + tid] = (float)cov[tid];
        if (a.cov64) a.cov64[(long long)b * 16 + tid] = cov[tid];
    }
    if (a.mask)
        for (int p = tid; p < P; p += NT) a.mask[(long long)b * P + p] = smask[p];
}

// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// K2: fused NOC-head post-processing.  One thread per RoI pixel; every read of all_pred is a coalesced
// row of the selected channel, every write a coalesced row of a channel-planar output map.
//   R9  flip/class channel pick   fcn_noc_decoder.py:225-267  (integer indexing, bit-exact)
//   R10 dim + NOC decode          multiclass_norm_dim_coder.py:28-36, noc_coder.py:50-73
//   R11 log-std decode            distance_invar_proj_error_coder.py:39-60 (distance=None)
//   R8  istd, RANSAC threshold    uncert_prop_pnp_optimizer.py:73,86-88
//   R12 RoI bin-centre grid       roi_align(coord_2d, ..., 'avg', aligned=True), interior analytic form
// fp32 with unfused multiply-adds, i.e. the rounding sequence of the reference's elementwise torch ops.
struct DecodeArgs {
    const float *all_pred; const long long *labels; const uint8_t *flip; const float *dim, *dim_var, *rois;
    int B, C, agnostic, h, w;
    const float *dim_means, *dim_stds; float noc_mean[3], noc_std[3];
    float k_epi, k_sd2, sd_sq, std_scale, ratio; int has_var;
    float *c2d, *istd, *c3d, *dims, *dims_var, *thr;
};

__global__ void __launch_bounds__(256) noc_decode_kernel(const DecodeArgs a) {
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    const int hw = a.h * a.w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int lab = (int)a.labels[b];
    const int c = a.agnostic ? 0 : lab;
    const int f = a.flip[b] ? 1 : 0;
    const int Cn = a.agnostic ? 1 : a.C;
    float dm[3], dv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float sd = a.dim_stds[lab * 3 + k];
        dm[k] = a.dim[b * 3 + k] * sd + a.dim_means[lab * 3 + k];
        dv[k] = a.has_var ? a.dim_var[b * 3 + k] * (sd * sd) : 0.0f;
    }
    const float x1 = a.rois[b * 4 + 0], y1 = a.rois[b * 4 + 1], x2 = a.rois[b * 4 + 2], y2 = a.rois[b * 4 + 3];
    const float su = (x2 - x1) / (float)a.w, sv = (y2 - y1) / (float)a.h;
    if (p == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (a.dims) a.dims[b * 3 + k] = dm[k];
            if (a.dims_var && a.has_var) a.dims_var[b * 3 + k] = dv[k];
        }
        if (a.thr) {
            const float v_last = (y1 - 0.5f) + ((float)(a.h - 1) + 0.5f) * sv, v_first = (y1 - 0.5f) + 0.5f * sv;
            a.thr[b] = a.ratio * (v_last - v_first);
        }
    }
    if (p >= hw) return;
    const int py = p / a.w, px = p - py * a.w;
    const float *base = a.all_pred + (long long)b * (2 * Cn * 5) * hw;
    const int ch_noc = f * 5 * Cn + 3 * c, ch_ls = f * 5 * Cn + 3 * Cn + 2 * c;
    float xv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float noc = base[(long long)(ch_noc + k) * hw + p];
        const float part = noc * a.noc_std[k] + a.noc_mean[k];
        a.c3d[((long long)b * 3 + k) * hw + p] = part * dm[k];
        xv[k] = dv[k] * (part * part);
    }
    const float v2[2] = { 0.5f * (xv[0] + xv[2]), xv[1] };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float ls = base[(long long)(ch_ls + k) * hw + p];
        float lspx;
        if (a.has_var) lspx = 0.5f * logf((v2[k] * a.k_epi + expf(2.0f * ls) * a.k_sd2) / a.sd_sq);
        else lspx = ls + 0.0f;                                    // log(sd / sd)
        a.istd[((long long)b * 2 + k) * hw + p] = expf(-lspx) / a.std_scale;
    }
    a.c2d[((long long)b * 2 + 0) * hw + p] = (x1 - 0.5f) + ((float)px + 0.5f) * su;
    a.c2d[((long long)b * 2 + 1) * hw + p] = (y1 - 0.5f) + ((float)py + 0.5f) * sv;
}

size_t lds_bytes(int Ppad, int wpo, size_t store_size) {
    size_t n = 0;
    n += sizeof(double) * wpo * kRedN;
    n += sizeof(unsigned long long) * ((Ppad / 64 + 1) & ~1);
    n += store_size * 7 * Ppad;
    n += sizeof(float) * kHyp * 8;
    n += sizeof(int) * wpo * kHyp;
    n += sizeof(float) * (2 * kMaxLeaves * 8 + 2 * kMaxLeaves + 2 * 16 + 2);
    n += sizeof(uint16_t) * Ppad;
    n += Ppad;
    return (n + 15) & ~(size_t)15;
}

void plan_rec(PairwisePlan &pl, int off, int n, bool &ok) {
    if (n <= 128) {
        if (pl.n_leaves >= kMaxLeaves) { ok = false; return; }
        pl.leaf_off[pl.n_leaves] = (uint16_t)off; pl.leaf_len[pl.n_leaves] = (uint16_t)n;
        pl.prog[pl.n_prog++] = (int8_t)pl.n_leaves++;
    } else {
        int n2 = n / 2; n2 -= n2 % 8;
        plan_rec(pl, off, n2, ok); if (!ok) return;
        plan_rec(pl, off + n2, n - n2, ok); if (!ok) return;
        pl.prog[pl.n_prog++] = -1;
    }
}

int g_last_hip_error = 0;
unsigned long long *g_stamps = nullptr;
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_last_hip_error = (int)e_; return MR_ERR_HIP; } } while (0)

template <typename T, int WPO>
int launch(const PnpArgs &a, hipStream_t st) {
    using S = typename Store<T>::type;
    const size_t lds = lds_bytes(a.Ppad, WPO, sizeof(S));
    if (lds > 160 * 1024) return MR_ERR_UNSUPPORTED;
    if (lds > 48 * 1024) {
        static std::mutex mu; static size_t granted = 0;
        std::lock_guard<std::mutex> lk(mu);
        if (lds > granted) {
            HIP_TRY(hipFuncSetAttribute((const void *)pnp_uncert_kernel<T, WPO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            granted = lds;
        }
    }
    hipLaunchKernelGGL((pnp_uncert_kernel<T, WPO>), dim3(a.B), dim3(64 * WPO), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

template <typename T>
int launch_wpo(PnpArgs &a, int wpo, hipStream_t st) {
    a.Ppad = ((a.P + 64 * wpo - 1) / (64 * wpo)) * (64 * wpo);
    switch (wpo) {
        case 1: return launch<T, 1>(a, st);
        case 2: return launch<T, 2>(a, st);
        case 4: return launch<T, 4>(a, st);
        case 8: return launch<T, 8>(a, st);
        default: return MR_ERR_BAD_ARGUMENT;
    }
}

int pick_wpo(int B, int P, int flags) {
    int w = (flags & MR_WAVES_MASK) >> MR_WAVES_SHIFT;
    if (w) return w;
    // fill 256 CUs x 4 SIMDs with ~4 waves per SIMD when the batch alone cannot
    w = 1;
    while (w < 8 && (long long)B * w * 2 <= 4096 && P >= 64 * w * 2) w *= 2;
    return w;
}

}  // namespace

// ================================================================================= C ABI =========
extern "C" {

int mr_pnp_version(void) { return MR_PNP_VERSION; }

const char *mr_pnp_error_string(int code) {
    switch (code) {
        case MR_OK: return "ok";
        case MR_ERR_BAD_ARGUMENT: return "bad argument";
        case MR_ERR_UNSUPPORTED: return "unsupported configuration (P too large for LDS, or unknown dtype)";
        case MR_ERR_HIP: return "HIP runtime error (see mr_pnp_last_hip_error)";
        case MR_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown error";
    }
}

int mr_pnp_last_hip_error(void) { return g_last_hip_error; }

// development aid (not in the public header): device buffer of (B,8) u64 cycle stamps, or NULL to disable
void mr_pnp_debug_set_stamps(unsigned long long *dev_ptr) { g_stamps = dev_ptr; }

int mr_pnp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mr_pnp_uncert_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    const float *ransac_thr, const double *init_pose, int B, int P,
    float z_min, float istd_thres, int inlier_opt_only, int flags,
    uint8_t *valid, float *pose, float *cov, float *tr_radius, uint8_t *inlier_mask, float *diag, void *stream) {
    if (B < 0 || P < 4 || P > 65535) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if (!x2d || !istd || !x3d || !x2d_strides || !istd_strides || !x3d_strides || !cam_mats || !u_range || !v_range ||
        !valid || !pose || !tr_radius || (!cov && !(flags & MR_COV_NONE)))
        return MR_ERR_BAD_ARGUMENT;
    if ((cam_batch != 1 && cam_batch != B) || (range_batch != 1 && range_batch != B)) return MR_ERR_BAD_ARGUMENT;
    PnpArgs a;
    memset(&a, 0, sizeof a);
    a.x2d = x2d; a.istd = istd; a.x3d = x3d;
    for (int i = 0; i < 3; ++i) { a.s2[i] = x2d_strides[i]; a.sw[i] = istd_strides[i]; a.s3[i] = x3d_strides[i]; }
    a.K = cam_mats; a.K_stride = (cam_batch == 1) ? 0 : 9; a.K_f64 = 0;
    a.ur = u_range; a.vr = v_range; a.r_stride = (range_batch == 1) ? 0 : 2; a.r_f64 = 0;
    a.ransac_thr = ransac_thr; a.init_pose = init_pose;
    a.B = B; a.P = P; a.z_min = (double)z_min; a.istd_thres = istd_thres; a.inlier_opt_only = inlier_opt_only; a.flags = flags;
    a.valid = valid; a.pose = pose; a.cov = cov; a.tr = tr_radius; a.mask = inlier_mask; a.diag = diag;
    a.stamps = g_stamps;
    int mm = flags & MR_MEAN_MASK;
    if (mm == MR_MEAN_AUTO) mm = (istd_strides[1] == 1 && P > 1) ? MR_MEAN_PAIRWISE : MR_MEAN_SEQUENTIAL;
    a.mean_mode = mm;
    if (mm == MR_MEAN_PAIRWISE && !(flags & MR_NO_ISTD_MASK)) {
        bool ok = true;
        plan_rec(a.plan, 0, P, ok);
        if (!ok) return MR_ERR_UNSUPPORTED;
    }
    const int wpo = pick_wpo(B, P, flags);
    hipStream_t st = (hipStream_t)stream;
    switch (in_dtype) {
        case MR_F32: return launch_wpo<float>(a, wpo, st);
        case MR_F16: return launch_wpo<__half>(a, wpo, st);
        case MR_F64: return launch_wpo<double>(a, wpo, st);
        default: return MR_ERR_UNSUPPORTED;
    }
}

int mr_noc_decode_batched(
    const float *all_pred, const int64_t *labels, const uint8_t *flip, const float *dim, const float *dim_var, const float *rois,
    int B, int num_classes, int class_agnostic, int h, int w,
    const float *dim_means, const float *dim_stds, const float *noc_means, const float *noc_stds,
    double proj_scaling_denominator, double ref_focal_y, double epistemic_std_gain, float std_scale, float ransac_thres_ratio,
    float *coords_2d, float *coords_2d_istd, float *coords_3d, float *dims, float *dims_var, float *ransac_thr, void *stream) {
    if (B < 0 || h < 1 || w < 1 || num_classes < 1 || B > 65535) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if (!all_pred || !labels || !flip || !dim || !rois || !dim_means || !dim_stds || !noc_means || !noc_stds ||
        !coords_2d || !coords_2d_istd || !coords_3d) return MR_ERR_BAD_ARGUMENT;
    DecodeArgs a;
    memset(&a, 0, sizeof a);
    a.all_pred = all_pred; a.labels = (const long long *)labels; a.flip = flip; a.dim = dim; a.dim_var = dim_var; a.rois = rois;
    a.B = B; a.C = num_classes; a.agnostic = class_agnostic; a.h = h; a.w = w;
    a.dim_means = dim_means; a.dim_stds = dim_stds;
    for (int k = 0; k < 3; ++k) { a.noc_mean[k] = noc_means[k]; a.noc_std[k] = noc_stds[k]; }
    // python-scalar constants of distance_invar_proj_error_coder.py:50-54, rounded the way torch rounds them
    const double e = ref_focal_y * epistemic_std_gain;
    a.k_epi = (float)(e * e);
    a.k_sd2 = (float)(proj_scaling_denominator * proj_scaling_denominator);
    const float sdf = (float)proj_scaling_denominator;
    a.sd_sq = sdf * sdf;
    a.std_scale = std_scale; a.ratio = ransac_thres_ratio; a.has_var = dim_var != nullptr;
    a.c2d = coords_2d; a.istd = coords_2d_istd; a.c3d = coords_3d; a.dims = dims; a.dims_var = dims_var;
    a.thr = (ransac_thres_ratio >= 0.f) ? ransac_thr : nullptr;
    const int hw = h * w;
    hipLaunchKernelGGL(noc_decode_kernel, dim3((hw + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

// The reference's per-object entry point (ext.h:1-13).  Host fp64 buffers; one object; blocking.
void pnp_uncert(double *pts2d, double *pts3d, double *wgt2d, double *K, double *init_pose,
                int *result_val, double *result_pose, double *result_cov, double *result_tr,
                int pn, double *clips) {
    *result_val = 0;
    memcpy(result_pose, init_pose, 4 * sizeof(double));                  // pnp_uncert_cpu.cpp:259
    *result_tr = 0.0;
    if (pn < 1 || pn > 65535) return;
    const int P = pn < 4 ? 4 : pn;
    // device staging: [pts2d 2P | pts3d 3P | wgt 2P | K 9 | ur 2 | vr 2 | init 4 | pose64 4 | cov64 16 | tr64 1] doubles
    //                 + [pose 4 | cov 16 | tr 1] floats + valid
    const size_t nd = (size_t)7 * P + 9 + 2 + 2 + 4 + 4 + 16 + 1;
    const size_t bytes = nd * sizeof(double) + 21 * sizeof(float) + 16;
    static std::mutex mu; static void *dbuf = nullptr; static size_t dcap = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (bytes > dcap) {
        if (dbuf) (void)hipFree(dbuf);
        dbuf = nullptr; dcap = 0;
        if (hipMalloc(&dbuf, bytes) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); return; }
        dcap = bytes;
    }
    std::vector<double> h(nd, 0.0);
    double *h2 = h.data(), *h3 = h2 + 2 * P, *hw = h3 + 3 * P, *hK = hw + 2 * P, *hur = hK + 9, *hvr = hur + 2, *hin = hvr + 2;
    memcpy(h2, pts2d, sizeof(double) * 2 * pn); memcpy(h3, pts3d, sizeof(double) * 3 * pn); memcpy(hw, wgt2d, sizeof(double) * 2 * pn);
    for (int p = pn; p < P; ++p) h3[3 * p + 2] = 1.0;                     // padded points carry zero weight
    memcpy(hK, K, sizeof(double) * 9);
    hur[0] = clips[1]; hur[1] = clips[2]; hvr[0] = clips[3]; hvr[1] = clips[4];
    memcpy(hin, init_pose, sizeof(double) * 4);
    if (hipMemcpy(dbuf, h.data(), nd * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); return; }
    double *d = (double *)dbuf;
    PnpArgs a;
    memset(&a, 0, sizeof a);
    a.x2d = d; a.x3d = d + 2 * P; a.istd = d + 5 * P;
    a.s2[0] = 0; a.s2[1] = 2; a.s2[2] = 1; a.sw[0] = 0; a.sw[1] = 2; a.sw[2] = 1; a.s3[0] = 0; a.s3[1] = 3; a.s3[2] = 1;
    a.K = d + 7 * P; a.K_stride = 0; a.K_f64 = 1;
    a.ur = d + 7 * P + 9; a.vr = d + 7 * P + 11; a.r_stride = 0; a.r_f64 = 1;
    a.init_pose = d + 7 * P + 13;
    a.pose64 = d + 7 * P + 17; a.cov64 = d + 7 * P + 21; a.tr64 = d + 7 * P + 37;
    float *df = (float *)(d + nd);
    a.pose = df; a.cov = df + 4; a.tr = df + 20; a.valid = (uint8_t *)(df + 21);
    a.B = 1; a.P = P; a.z_min = clips[0]; a.istd_thres = 0.f; a.inlier_opt_only = 0;
    a.flags = MR_NO_ISTD_MASK | (result_cov ? MR_COV_CERES : MR_COV_NONE);
    a.mean_mode = MR_MEAN_SEQUENTIAL;
    int wpo = 1; while (wpo < 8 && P >= 64 * wpo * 2) wpo *= 2;
    if (launch_wpo<double>(a, wpo, nullptr) != MR_OK) return;
    std::vector<double> ho(21); uint8_t hv[16];
    if (hipMemcpy(ho.data(), a.pose64, 21 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(hv, a.valid, 16, hipMemcpyDeviceToHost) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); return; }
    memcpy(result_pose, ho.data(), 4 * sizeof(double));
    *result_tr = ho[20];
    *result_val = hv[0] ? 1 : 0;
    if (hv[0] && result_cov) memcpy(result_cov, ho.data() + 4, 16 * sizeof(double));
}

}  // extern "C"

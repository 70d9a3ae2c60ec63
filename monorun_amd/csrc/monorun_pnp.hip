// monorun_pnp.hip — gfx950 (MI355X / CDNA4) kernels + C ABI for MonoRUn's uncertainty-aware PnP hot path.
// The C ABI is declared in include/monorun_pnp.h; the fused per-object kernel lives in pnp_kernel.inc.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <map>
#include <utility>
#include <stdlib.h>
#include <vector>
#include <type_traits>

#include "monorun_pnp.h"

namespace {

constexpr int kMaxLeaves = 64;      // numpy pairwise-sum leaves (blocks of <=128) supported per object
constexpr int kMaxChunks = 128;     // 64-point chunks per object (P <= 8192; LDS caps P well below that)
constexpr int kHyp = 32;            // K0 hypotheses (the reference's RANSAC runs 30 iterations)
constexpr uint32_t kK0Seed = 0x9E3779B9u;
constexpr int kRedN = 32;           // doubles per wave in the cross-wave reduction scratch
constexpr int kMaxDevices = 64;     // per-device library state (LDS opt-in, host-entry staging) is keyed by HIP device id
#ifndef MR_MIN_WAVES
#define MR_MIN_WAVES 3              // waves per SIMD the register allocator must allow: fp64 storage (<= 168 VGPRs)
#endif
#ifndef MR_RELAXED_WPO
#define MR_RELAXED_WPO 2            // instantiations with at most this many waves per object are compiled for 3 waves per SIMD (168 VGPRs):
                                    // they serve large batches, where LDS allows 5 workgroups = 10 waves per CU anyway; 8192-object launch 297 -> 289 us
#endif
#ifndef MR_MIN_WAVES_F32
#define MR_MIN_WAVES_F32 4          // fp32 and 16-bit storage (the pipeline's cases): <= 128 VGPRs, so that the 1024 four-wave blocks of a
#endif                              // config-2 launch are all resident (measured: 76 us vs 83 us when the allocator lands on 139 VGPRs)

// ------------------------------------------------------------------------------------------------
// K2: fused NOC-head post-processing.  One thread per RoI pixel; every read of all_pred is a coalesced
// row of the selected channel, every write a coalesced row of a channel-planar output map.
//   R9  flip/class channel pick   fcn_noc_decoder.py:225-267  (integer indexing, bit-exact)
//   R10 dim + NOC decode          multiclass_norm_dim_coder.py:28-36, noc_coder.py:50-73
//   R11 log-std decode            distance_invar_proj_error_coder.py:39-60 (distance=None)
//   R8  istd, RANSAC threshold    uncert_prop_pnp_optimizer.py:73,86-88
//   R12 RoI bin-centre grid       roi_align(coord_2d, ..., 'avg', aligned=True), interior analytic form
// fp32 with unfused multiply-adds, i.e. the rounding sequence of the reference's elementwise torch ops.
// The per-object / per-pixel arithmetic is shared with the fused path of the PnP kernel (decoded maps
// written straight into its LDS tile, never to HBM).
struct DecodeArgs {
    const void *all_pred; int pred_dtype;      // head output: MR_F32, MR_F16 or MR_BF16 (autocast pipelines); decoded in fp32
    const long long *labels; const uint8_t *flip; const float *dim, *dim_var, *rois;
    int B, C, agnostic, h, w;
    const float *dim_means, *dim_stds, *noc_means, *noc_stds;     // device pointers: (C,3), (C,3), (3), (3)
    float k_epi, k_sd2, sd_sq, std_scale, ratio; int has_var;
    float *c2d, *istd, *c3d, *dims, *dims_var, *thr;
    const float *map2d; int map_h, map_w;      // optional coord_2d map (2, H, W): exact RoIAlign sampling instead of the analytic grid
    unsigned w_magic;                          // floor((2^32 - 1) / w) + 1 (0 when w == 1 or h * w >= 65536): row index p / w == __umulhi(p, w_magic), decode_pixel_pair
};

// RoIAlign forward, average pooling (mmcv.ops.roi_align: the published Detectron/mmcv algorithm, mmcv 1.2.1
// roi_align_cuda_kernel.cuh — third-party, not in the reference tree): bilinear taps with mmcv's border rules
// (a sample more than one pixel outside contributes 0; otherwise it is clamped into [0, size-1]).
__device__ __forceinline__ float roi_bilinear(const float *in, int H, int W, float y, float x) {
#pragma clang fp contract(off)
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.0f - ly, hx = 1.0f - lx;
    const float v1 = in[y_low * W + x_low], v2 = in[y_low * W + x_high], v3 = in[y_high * W + x_low], v4 = in[y_high * W + x_high];
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

// one output bin (ph, pw) of one channel; roi = x1 y1 x2 y2 already multiplied by spatial_scale
__device__ __forceinline__ float roi_align_avg_bin(const float *in, int H, int W, float x1, float y1, float x2, float y2,
                                                   int ph, int pw, int out_h, int out_w, int sampling_ratio, int aligned) {
#pragma clang fp contract(off)
    const float off = aligned ? 0.5f : 0.0f;
    const float sw = x1 - off, sh = y1 - off;
    float rw = (x2 - off) - sw, rh = (y2 - off) - sh;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    const float bh = rh / (float)out_h, bw = rw / (float)out_w;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)out_h);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)out_w);
    const float count = (float)max(gh * gw, 1);
    float acc = 0.0f;
    for (int iy = 0; iy < gh; ++iy) {
        const float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
            const float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
            acc += roi_bilinear(in, H, W, y, x);
        }
    }
    return acc / count;
}

__global__ void __launch_bounds__(256) roi_align_avg_kernel(const float *in, const float *rois, int K, int C, int H, int W, int out_h, int out_w,
                                                            float spatial_scale, int sampling_ratio, int aligned, float *out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)K * C * out_h * out_w) return;
    const int pw = (int)(idx % out_w), ph = (int)((idx / out_w) % out_h), c = (int)((idx / ((long long)out_w * out_h)) % C);
    const int n = (int)(idx / ((long long)out_w * out_h * C));
    const float *r = rois + (long long)n * 5;
    const int bi = (int)r[0];
    out[idx] = roi_align_avg_bin(in + ((long long)bi * C + c) * H * W, H, W, r[1] * spatial_scale, r[2] * spatial_scale, r[3] * spatial_scale,
                                 r[4] * spatial_scale, ph, pw, out_h, out_w, sampling_ratio, aligned);
}

// exp / log of the istd chain, SPECIFIED (not library calls): the decoded istd feeds a bit-exact threshold (the istd inlier
// mask, pnp_uncert_cpu.py:164-168), so its last bit must not depend on which libm / device library computed it.  Classical
// single-precision algorithms (Cephes expf / logf: Cody-Waite reduction with the two-part ln 2, degree-5 / degree-8
// polynomials) written as a fixed sequence of IEEE float32 multiplications and additions — no fma, no contraction — which the
// test infrastructure restates operation for operation with numpy float32 arithmetic (spec_expf / spec_logf there).  Error <= 1 ulp.
__device__ __forceinline__ float mr_expf(float x) {
#pragma clang fp contract(off)
    if (x > 88.72283935546875f) return __int_as_float(0x7f800000);
    if (x < -103.0f) return 0.0f;
    const float kf = rintf(x * 1.44269504088896341f);
    float r = x - kf * 0.693359375f;
    r = r - kf * -2.12194440e-4f;
    const float z = r * r;
    float p = 1.9875691500E-4f * r + 1.3981999507E-3f;
    p = p * r + 8.3334519073E-3f;
    p = p * r + 4.1665795894E-2f;
    p = p * r + 1.6666665459E-1f;
    p = p * r + 5.0000001201E-1f;
    float y = p * z + r;
    y = y + 1.0f;
    return ldexpf(y, (int)kf);                 // NaN in -> NaN out (both range tests are false)
}
__device__ __forceinline__ float mr_logf(float x) {
#pragma clang fp contract(off)
    if (!(x > 0.0f)) return x == 0.0f ? -__int_as_float(0x7f800000) : __int_as_float(0x7fc00000);
    if (x == __int_as_float(0x7f800000)) return x;
    int e;
    float m = frexpf(x, &e);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else m = m - 1.0f;
    const float z = m * m;
    float p = 7.0376836292E-2f * m - 1.1514610310E-1f;
    p = p * m + 1.1676998740E-1f;
    p = p * m - 1.2420140846E-1f;
    p = p * m + 1.4249322787E-1f;
    p = p * m - 1.6668057665E-1f;
    p = p * m + 2.0000714765E-1f;
    p = p * m - 2.4999993993E-1f;
    p = p * m + 3.3333331174E-1f;
    const float fe = (float)e;
    float y = m * (z * p);
    y = y + -2.12194440e-4f * fe;
    y = y - 0.5f * z;
    const float zz = m + y;
    return zz + 0.693359375f * fe;
}

struct DecodeObj { float dm[3], dv[3], nm[3], ns[3]; float x1, y1, x2, y2, su, sv, thr; long long base; int ch_noc, ch_ls; };   // base: element offset of the object

__device__ __forceinline__ float pred_at(const DecodeArgs &a, long long i) {
    if (a.pred_dtype == MR_F32) return ((const float *)a.all_pred)[i];
    if (a.pred_dtype == MR_F16) return __half2float(((const __half *)a.all_pred)[i]);
    return __uint_as_float((unsigned)((const unsigned short *)a.all_pred)[i] << 16);            // bfloat16
}

__device__ __forceinline__ void decode_object(const DecodeArgs &a, int b, DecodeObj &o) {
#pragma clang fp contract(off)
    const int hw = a.h * a.w;
    const int lab = (int)a.labels[b];
    const int c = a.agnostic ? 0 : lab;
    const int f = a.flip[b] ? 1 : 0;
    const int Cn = a.agnostic ? 1 : a.C;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float sd = a.dim_stds[lab * 3 + k];
        o.dm[k] = a.dim[b * 3 + k] * sd + a.dim_means[lab * 3 + k];
        o.dv[k] = a.has_var ? a.dim_var[b * 3 + k] * (sd * sd) : 0.0f;
        o.nm[k] = a.noc_means[k]; o.ns[k] = a.noc_stds[k];
    }
    const float x1 = a.rois[b * 4 + 0], y1 = a.rois[b * 4 + 1], x2 = a.rois[b * 4 + 2], y2 = a.rois[b * 4 + 3];
    o.x1 = x1; o.y1 = y1;
    o.su = (x2 - x1) / (float)a.w; o.sv = (y2 - y1) / (float)a.h;
    o.x2 = x2; o.y2 = y2;
    float v_last, v_first;
    if (a.map2d) {      // x2d[:, 1, -1, 0] - x2d[:, 1, 0, 0] of the sampled map (uncert_prop_pnp_optimizer.py:86-88)
        const float *mv = a.map2d + (long long)a.map_h * a.map_w;
        v_last = roi_align_avg_bin(mv, a.map_h, a.map_w, x1, y1, x2, y2, a.h - 1, 0, a.h, a.w, 0, 1);
        v_first = roi_align_avg_bin(mv, a.map_h, a.map_w, x1, y1, x2, y2, 0, 0, a.h, a.w, 0, 1);
    } else {
        v_last = (y1 - 0.5f) + ((float)(a.h - 1) + 0.5f) * o.sv; v_first = (y1 - 0.5f) + 0.5f * o.sv;
    }
    o.thr = a.ratio * (v_last - v_first);
    o.base = (long long)b * (2 * Cn * 5) * hw;
    o.ch_noc = f * 5 * Cn + 3 * c; o.ch_ls = f * 5 * Cn + 3 * Cn + 2 * c;
}

// (the scalar form, textually what the fused PnP kernel has been tuned around: its code must not move — tools/isa_diff.sh)
__device__ __forceinline__ void decode_pixel(const DecodeArgs &a, const DecodeObj &o, int p, float (&c2d)[2], float (&istd)[2], float (&c3d)[3]) {
#pragma clang fp contract(off)
    const int hw = a.h * a.w;
    const int py = p / a.w, px = p - py * a.w;
    float xv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float noc = pred_at(a, o.base + (long long)(o.ch_noc + k) * hw + p);
        const float part = noc * o.ns[k] + o.nm[k];
        c3d[k] = part * o.dm[k];
        xv[k] = o.dv[k] * (part * part);
    }
    const float v2[2] = { 0.5f * (xv[0] + xv[2]), xv[1] };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float ls = pred_at(a, o.base + (long long)(o.ch_ls + k) * hw + p);
        float lspx;
        if (a.has_var) lspx = 0.5f * mr_logf((v2[k] * a.k_epi + mr_expf(2.0f * ls) * a.k_sd2) / a.sd_sq);
        else lspx = ls + 0.0f;                                    // log(sd / sd)
        istd[k] = mr_expf(-lspx) / a.std_scale;
    }
    if (a.map2d) {      // roi_align(coord_2d, rois, (h, w), 1.0, 0, 'avg', True)   (monorun_roi_head.py:521-523)
        c2d[0] = roi_align_avg_bin(a.map2d, a.map_h, a.map_w, o.x1, o.y1, o.x2, o.y2, py, px, a.h, a.w, 0, 1);
        c2d[1] = roi_align_avg_bin(a.map2d + (long long)a.map_h * a.map_w, a.map_h, a.map_w, o.x1, o.y1, o.x2, o.y2, py, px, a.h, a.w, 0, 1);
    } else {            // interior analytic form: the bin centre of an identity coordinate map
        c2d[0] = (o.x1 - 0.5f) + ((float)px + 0.5f) * o.su;
        c2d[1] = (o.y1 - 0.5f) + ((float)py + 0.5f) * o.sv;
    }
}

// the same arithmetic from five head-channel VALUES (the vector kernel loads them four pixels at a time); analytic grid only
__device__ __forceinline__ void decode_pixel_vals(const DecodeArgs &a, const DecodeObj &o, int p, const float (&nocv)[3], const float (&lsv)[2],
                                                  float (&c2d)[2], float (&istd)[2], float (&c3d)[3]) {
#pragma clang fp contract(off)
    const int py = p / a.w, px = p - py * a.w;
    float xv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float part = nocv[k] * o.ns[k] + o.nm[k];
        c3d[k] = part * o.dm[k];
        xv[k] = o.dv[k] * (part * part);
    }
    const float v2[2] = { 0.5f * (xv[0] + xv[2]), xv[1] };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float ls = lsv[k];
        float lspx;
        if (a.has_var) lspx = 0.5f * mr_logf((v2[k] * a.k_epi + mr_expf(2.0f * ls) * a.k_sd2) / a.sd_sq);
        else lspx = ls + 0.0f;                                    // log(sd / sd)
        istd[k] = mr_expf(-lspx) / a.std_scale;
    }
    if (a.map2d) {      // roi_align(coord_2d, rois, (h, w), 1.0, 0, 'avg', True)   (monorun_roi_head.py:521-523)
        c2d[0] = roi_align_avg_bin(a.map2d, a.map_h, a.map_w, o.x1, o.y1, o.x2, o.y2, py, px, a.h, a.w, 0, 1);
        c2d[1] = roi_align_avg_bin(a.map2d + (long long)a.map_h * a.map_w, a.map_h, a.map_w, o.x1, o.y1, o.x2, o.y2, py, px, a.h, a.w, 0, 1);
    } else {            // interior analytic form: the bin centre of an identity coordinate map
        c2d[0] = (o.x1 - 0.5f) + ((float)px + 0.5f) * o.su;
        c2d[1] = (o.y1 - 0.5f) + ((float)py + 0.5f) * o.sv;
    }
}

__global__ void __launch_bounds__(256) noc_decode_kernel(const DecodeArgs a) {
    const int hw = a.h * a.w;
    const int bpo = (hw + 255) >> 8;                   // blocks per object; 1-D grid: B * bpo <= 2^31 - 1
    const int b = blockIdx.x / bpo;
    const int p = (blockIdx.x - b * bpo) * 256 + threadIdx.x;
    DecodeObj o;
    decode_object(a, b, o);
    if (p == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (a.dims) a.dims[b * 3 + k] = o.dm[k];
            if (a.dims_var && a.has_var) a.dims_var[b * 3 + k] = o.dv[k];
        }
        if (a.thr) a.thr[b] = o.thr;
    }
    if (p >= hw) return;
    float c2d[2], istd[2], c3d[3];
    decode_pixel(a, o, p, c2d, istd, c3d);
#pragma unroll
    for (int k = 0; k < 3; ++k) a.c3d[((long long)b * 3 + k) * hw + p] = c3d[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) { a.istd[((long long)b * 2 + k) * hw + p] = istd[k]; a.c2d[((long long)b * 2 + k) * hw + p] = c2d[k]; }
}

// Two-pixel forms of the same arithmetic for the vector kernel: every multiplication and addition of the specified sequences acts on
// a PAIR of pixels (v_pk_mul_f32 / v_pk_add_f32: one instruction, two IEEE float32 results, each bit-identical to the scalar
// operation), the special cases become selects after the common path.  mr_expf / mr_logf / decode_pixel_vals stay the definition.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// CHECKED = false: the common path only — the caller is told (`special`) when an argument falls into a special case and redoes the
// work with the checked form; on ordinary inputs this drops the compares and selects (~ 10 % of the vector kernel's instructions).
template <bool CHECKED = true>
__device__ __forceinline__ f32x2 mr_expf2(f32x2 x, bool *special = nullptr) {
#pragma clang fp contract(off)
    f32x2 kf;
    kf.x = rintf(x.x * 1.44269504088896341f); kf.y = rintf(x.y * 1.44269504088896341f);
    f32x2 r = x - kf * 0.693359375f;
    r = r - kf * -2.12194440e-4f;
    const f32x2 z = r * r;
    f32x2 p = 1.9875691500E-4f * r + 1.3981999507E-3f;
    p = p * r + 8.3334519073E-3f;
    p = p * r + 4.1665795894E-2f;
    p = p * r + 1.6666665459E-1f;
    p = p * r + 5.0000001201E-1f;
    f32x2 y = p * z + r;
    y = y + 1.0f;
    f32x2 o;
    o.x = ldexpf(y.x, (int)kf.x); o.y = ldexpf(y.y, (int)kf.y);
    if constexpr (CHECKED) {
        o.x = x.x > 88.72283935546875f ? __int_as_float(0x7f800000) : (x.x < -103.0f ? 0.0f : o.x);
        o.y = x.y > 88.72283935546875f ? __int_as_float(0x7f800000) : (x.y < -103.0f ? 0.0f : o.y);
    } else {
        // conservative: |x + 7.14| > 95.8 holds for every x > 88.72283935546875 and every x < -103 (and for a sliver inside the range: a
        // false alarm only costs the redo); a NaN is not special — it goes through the same arithmetic in the checked form
        *special = *special || (fabsf(x.x + 7.14f) > 95.8f) || (fabsf(x.y + 7.14f) > 95.8f);
    }
    return o;
}
template <bool CHECKED = true>
__device__ __forceinline__ f32x2 mr_logf2(f32x2 x, bool *special = nullptr) {
#pragma clang fp contract(off)
    int e0, e1;
    f32x2 m;
    m.x = frexpf(x.x, &e0); m.y = frexpf(x.y, &e1);
    const bool lo0 = m.x < 0.707106781186547524f, lo1 = m.y < 0.707106781186547524f;
    e0 -= lo0 ? 1 : 0; e1 -= lo1 ? 1 : 0;
    const f32x2 m2 = m + m - 1.0f, m1 = m - 1.0f;
    m.x = lo0 ? m2.x : m1.x; m.y = lo1 ? m2.y : m1.y;
    const f32x2 z = m * m;
    f32x2 p = 7.0376836292E-2f * m - 1.1514610310E-1f;
    p = p * m + 1.1676998740E-1f;
    p = p * m - 1.2420140846E-1f;
    p = p * m + 1.4249322787E-1f;
    p = p * m - 1.6668057665E-1f;
    p = p * m + 2.0000714765E-1f;
    p = p * m - 2.4999993993E-1f;
    p = p * m + 3.3333331174E-1f;
    f32x2 fe;
    fe.x = (float)e0; fe.y = (float)e1;
    f32x2 y = m * (z * p);
    y = y + -2.12194440e-4f * fe;
    y = y - 0.5f * z;
    const f32x2 zz = m + y;
    f32x2 o = zz + 0.693359375f * fe;
    const float inf = __int_as_float(0x7f800000), nan = __int_as_float(0x7fc00000);
    if constexpr (CHECKED) {
        o.x = !(x.x > 0.0f) ? (x.x == 0.0f ? -inf : nan) : (x.x == inf ? x.x : o.x);
        o.y = !(x.y > 0.0f) ? (x.y == 0.0f ? -inf : nan) : (x.y == inf ? x.y : o.y);
    } else {
        *special = *special || !(x.x > 0.0f) || !(x.y > 0.0f) || x.x == inf || x.y == inf;      // zero, negative, NaN, +inf
    }
    return o;
}
// pixels p and p + 1 of one object row-major (p even, same row: w is even whenever h * w % 4 == 0 ... not required: px / py per pixel)
// x / c for a wave-uniform float32 c, correctly rounded like the IEEE division it replaces, in 3 instructions instead of ~12: the
// quotient is formed in float64 as x * RN64(1 / c) (relative error < 2^-52) and rounded to float32 once.  A float32 quotient of two
// float32 numbers is never closer than 2^-49 (relative) to a rounding boundary — with X, C the 24-bit significands and M the odd 25-bit
// significand of a midpoint, X 2^s - M C is a non-zero integer — so that single rounding lands on the IEEE result; zeros, infinities,
// NaNs, c = 0, overflow and float32 denormals go through the float64 product and the conversion unchanged.  rc = 1.0 / (double)c.
__device__ __forceinline__ float div_by_uniform(float x, double rc) { return (float)((double)x * rc); }

// w_magic = floor((2^32 - 1) / w) + 1: p / w == __umulhi(p, w_magic) for p, w < 2^16 (the error of the product is p (w_magic w - 2^32)
// / (w 2^32) < p / 2^32 < 1 / w) — the two integer divisions per pixel pair were ~ 12 % of the kernel's instructions
// The two pixels are p and p + 1 (the vector decode kernel), or p and pb (the fused kernel's load stage: a lane's pixels are a stride apart).
// CHECKED = false needs `special` (it is written); CHECKED = true ignores it.
template <bool CHECKED = true, bool ADJACENT = true>
__device__ __forceinline__ void decode_pixel_pair(const DecodeArgs &a, const DecodeObj &o, int p, const f32x2 (&nocv)[3], const f32x2 (&lsv)[2],
                                                  f32x2 (&c2d)[2], f32x2 (&istd)[2], f32x2 (&c3d)[3], double rc_sd_sq, double rc_std_scale, unsigned w_magic,
                                                  bool *special = nullptr, int pb = 0) {
#pragma clang fp contract(off)
    f32x2 xv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const f32x2 part = nocv[k] * o.ns[k] + o.nm[k];
        c3d[k] = part * o.dm[k];
        xv[k] = o.dv[k] * (part * part);
    }
    const f32x2 v2[2] = { 0.5f * (xv[0] + xv[2]), xv[1] };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const f32x2 ls = lsv[k];
        f32x2 lspx;
        if (a.has_var) {
            const f32x2 num = v2[k] * a.k_epi + mr_expf2<CHECKED>(2.0f * ls, special) * a.k_sd2;
            f32x2 q;
            q.x = div_by_uniform(num.x, rc_sd_sq); q.y = div_by_uniform(num.y, rc_sd_sq);
            lspx = 0.5f * mr_logf2<CHECKED>(q, special);
        } else lspx = ls + 0.0f;                                  // log(sd / sd)
        const f32x2 ex = mr_expf2<CHECKED>(-lspx, special);
        istd[k].x = div_by_uniform(ex.x, rc_std_scale); istd[k].y = div_by_uniform(ex.y, rc_std_scale);
    }
    const int py0 = (int)__umulhi((unsigned)p, w_magic), px0 = p - py0 * a.w;
    int py1, px1;
    if constexpr (ADJACENT) {
        const bool wrap = px0 + 1 == a.w;                         // pixel p + 1 starts the next row
        py1 = wrap ? py0 + 1 : py0; px1 = wrap ? 0 : px0 + 1;
    } else { py1 = (int)__umulhi((unsigned)pb, w_magic); px1 = pb - py1 * a.w; }
    f32x2 fx, fy;
    fx.x = (float)px0; fx.y = (float)px1; fy.x = (float)py0; fy.y = (float)py1;
    c2d[0] = (o.x1 - 0.5f) + (fx + 0.5f) * o.su;
    c2d[1] = (o.y1 - 0.5f) + (fy + 0.5f) * o.sv;
}

// K2, vector form: one thread per FOUR consecutive RoI pixels of one object — five 16-byte loads of the selected head channels,
// seven 16-byte NON-TEMPORAL stores of the decoded channels.  One workgroup per object (grid = B).  Same per-pixel arithmetic as the
// scalar kernel above (two pixels per packed instruction), hence bit-identical outputs.  What bounds it, as measured (per-wave 100 MHz
// stamps of a -DMR_K2_EXPERIMENT build, tools/gpu_k2_timeline.py, profiles/r04_k2_timeline.txt, r04_k2_store_policy.txt):
//   * a wave has its parameters 2.0 us after it starts (three dependent rounds of loads), its pixel loads out 0.9 us later, the data
//     0.5 us later — the pixel data is NOT what is late —, its arithmetic done after another 3.1 us; waves start within 0.7 us;
//   * with plain stores the profiler counted 2 - 3 us more than the last wave's end: the write-back of the 22 MB of outputs from L2 when
//     the dispatch ends.  Non-temporal stores send them on during the launch (12.1 - 12.7 -> 10.2 - 11.9 us);
//   * from there the launch is instruction-issue-bound, and cuts of the stream count: the object's last wave on pixel pairs, the row
//     index by multiplication, the exp / log sequences on their common path first (809 -> 620 VALU instructions per wave; 9.5 - 9.8 us
//     per launch issued back to back = 0.49 - 0.51 of 8 TB/s, an isolated launch 8.0 us).
// Measured and not kept (same files; HISTORY.md): a persistent software-pipelined form, 128- and 64-thread workgroups, caps on the
// resident workgroups, a grouped form (several objects per workgroup, lanes numbered through their quads), non-temporal loads, the class
// rows fetched ahead of the label, the pixel loads issued ahead of the other parameters, starting the waves of a SIMD apart, Horner
// chains interleaved across four pixels.  Requires fp32 head output, h * w % 4 == 0 and < 65536, no coord_2d map (the launcher falls
// back to the scalar kernel otherwise).
template <int THREADS, int TRIPS>
__global__ void __launch_bounds__(THREADS) noc_decode_kernel_x4(const DecodeArgs a, int quads_per_obj
#ifdef MR_K2_EXPERIMENT
    , unsigned long long *stamps
#endif
    ) {
#ifdef MR_K2_EXPERIMENT
#define K2_STAMP(i) do { if (stamps && (threadIdx.x & 63) == 0) stamps[((long long)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); } while (0)
    K2_STAMP(0);
#else
#define K2_STAMP(i) do { } while (0)
#endif
    // one workgroup per object: the object index is wave-uniform, so its parameters (label, flip, dims, RoI, coder constants — two
    // dependent rounds of loads) are fetched through the scalar cache once per wave instead of once per lane.  A thread takes up
    // to TRIPS pixel quads (q = t, t + THREADS, ...): all their loads are issued before the first quad is decoded, so the
    // arithmetic of one quad (the specified exp / log sequences and IEEE divisions: ~300 instructions per pixel) overlaps the
    // loads of the next and the stores of the previous one.
    const int b = blockIdx.x;
    const int hw = a.h * a.w;
    const unsigned w_magic = a.w_magic;
    DecodeObj o;
    decode_object(a, b, o);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (a.dims) a.dims[b * 3 + k] = o.dm[k];
            if (a.dims_var && a.has_var) a.dims_var[b * 3 + k] = o.dv[k];
        }
        if (a.thr) a.thr[b] = o.thr;
    }
    const float *ap = (const float *)a.all_pred;
    const double rc_sd_sq = 1.0 / (double)a.sd_sq, rc_std_scale = 1.0 / (double)a.std_scale;      // div_by_uniform
    K2_STAMP(1);
    if constexpr (TRIPS == 1) {
        // The last wave of an object owns only the quads left over (28x28: 4 of 196) and would still issue the whole two-pairs-per-lane
        // instruction stream for them.  With at most 32 quads left it works on PAIRS instead: lane l takes pixels (2l, 2l + 1) of the
        // wave's range — one pass through the same packed arithmetic, half the instructions, 8-byte loads and stores.
        const int w0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 64;      // first quad of this wave
        const int nq = quads_per_obj - w0;
        if (nq > 0 && nq <= 32) {
            const int l = threadIdx.x & 63;
            if (l < 2 * nq) {
                const int p0 = 4 * w0 + 2 * l;
                typedef float f32x2v __attribute__((ext_vector_type(2)));
                f32x2 noc[3], ls[2], c2[2], w2[2], c3[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { const float2 v = *(const float2 *)(ap + o.base + (long long)(o.ch_noc + k) * hw + p0); noc[k].x = v.x; noc[k].y = v.y; }
#pragma unroll
                for (int k = 0; k < 2; ++k) { const float2 v = *(const float2 *)(ap + o.base + (long long)(o.ch_ls + k) * hw + p0); ls[k].x = v.x; ls[k].y = v.y; }
                bool special = false;
                decode_pixel_pair<false>(a, o, p0, noc, ls, c2, w2, c3, rc_sd_sq, rc_std_scale, w_magic, &special);
                if (special) decode_pixel_pair<true>(a, o, p0, noc, ls, c2, w2, c3, rc_sd_sq, rc_std_scale, w_magic);
                auto st2 = [](float *dst, f32x2 v) { __builtin_nontemporal_store(f32x2v{ v.x, v.y }, (f32x2v *)dst); };
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    st2(a.c2d + ((long long)b * 2 + k) * hw + p0, c2[k]);
                    st2(a.istd + ((long long)b * 2 + k) * hw + p0, w2[k]);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) st2(a.c3d + ((long long)b * 3 + k) * hw + p0, c3[k]);
            }
            return;
        }
    }
    for (int q0 = threadIdx.x; q0 < quads_per_obj; q0 += THREADS * TRIPS) {
        float4 in[TRIPS][5];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int q = q0 + t * THREADS;
            if (q < quads_per_obj) {
#pragma unroll
                for (int k = 0; k < 3; ++k) in[t][k] = *(const float4 *)(ap + o.base + (long long)(o.ch_noc + k) * hw + 4 * q);
#pragma unroll
                for (int k = 0; k < 2; ++k) in[t][3 + k] = *(const float4 *)(ap + o.base + (long long)(o.ch_ls + k) * hw + 4 * q);
            }
        }
#ifdef MR_K2_EXPERIMENT
        K2_STAMP(2);
        if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); K2_STAMP(3); }
#endif
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int q = q0 + t * THREADS;
            if (q >= quads_per_obj) break;
            const int p0 = 4 * q;
            float out[7][4];
            // pixel pairs (p0, p0 + 1), (p0 + 2, p0 + 3): packed float32 arithmetic.  First the common path of the specified exp / log
            // sequences (no range tests, no selects); a lane that met a special input redoes its quad with the checked forms.
            auto quad = [&](auto checked, bool *special) {
                constexpr bool CHECKED = decltype(checked)::value;
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    f32x2 noc[3], ls[2], c2[2], w2[2], c3[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { noc[k].x = ((const float *)&in[t][k])[j]; noc[k].y = ((const float *)&in[t][k])[j + 1]; }
#pragma unroll
                    for (int k = 0; k < 2; ++k) { ls[k].x = ((const float *)&in[t][3 + k])[j]; ls[k].y = ((const float *)&in[t][3 + k])[j + 1]; }
#ifdef MR_K2_COPY_ONLY      // ubench: the kernel's memory traffic without its arithmetic (tools/profile_k2_quick.sh on a variant build)
                    c2[0] = noc[0]; c2[1] = noc[1]; w2[0] = ls[0]; w2[1] = ls[1]; c3[0] = noc[2]; c3[1] = noc[0] + ls[0]; c3[2] = noc[1] + ls[1];
#else
                    decode_pixel_pair<CHECKED>(a, o, p0 + j, noc, ls, c2, w2, c3, rc_sd_sq, rc_std_scale, w_magic, special);
#endif
                    out[0][j] = c2[0].x; out[0][j + 1] = c2[0].y; out[1][j] = c2[1].x; out[1][j + 1] = c2[1].y;
                    out[2][j] = w2[0].x; out[2][j + 1] = w2[0].y; out[3][j] = w2[1].x; out[3][j + 1] = w2[1].y;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { out[4 + k][j] = c3[k].x; out[4 + k][j + 1] = c3[k].y; }
                }
            };
            bool special = false;
            quad(std::false_type{}, &special);
            if (special) quad(std::true_type{}, nullptr);
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            // NON-TEMPORAL stores (`global_store_dwordx4 ... nt`): the 22 MB a launch writes are not kept in L2, so the write-back at the end
            // of the dispatch has little left to do: 10.2 - 11.3 us per launch against 12.1 - 12.7 us with plain stores (300 launches each way,
            // alternating; `sc0 sc1` write-through stores give the same, non-temporal LOADS nothing: profiles/r04_k2_store_policy.txt)
            auto st4 = [](float *dst, const float (&v)[4]) { __builtin_nontemporal_store(f32x4{ v[0], v[1], v[2], v[3] }, (f32x4 *)dst); };
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                st4(a.c2d + ((long long)b * 2 + k) * hw + p0, out[k]);
                st4(a.istd + ((long long)b * 2 + k) * hw + p0, out[2 + k]);
            }
            K2_STAMP(4);
#pragma unroll
            for (int k = 0; k < 3; ++k) st4(a.c3d + ((long long)b * 3 + k) * hw + p0, out[4 + k]);
        }
    }
#ifdef MR_K2_EXPERIMENT
    K2_STAMP(5);
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); K2_STAMP(6); }
#endif
#undef K2_STAMP
}

// numpy's pairwise summation tree for a length-P contiguous float32 reduction, built on the host:
// leaves (blocks of <=128 elements) + the combine tree, internal nodes ordered by height so that the
// kernel can evaluate it level by level.  Value slots: [0, n_leaves) leaves, n_leaves + k internal k.
struct PairwisePlan {
    int n_leaves, n_internal, n_levels, root;
    uint16_t leaf_off[kMaxLeaves];
    uint16_t leaf_len[kMaxLeaves];
    uint8_t left[kMaxLeaves], right[kMaxLeaves];
    uint8_t level_start[16];
};

struct PnpArgs {
    const void *x2d, *istd, *x3d;
    long long s2[3], sw[3], s3[3];            // global element strides (b, p, c)
    int vec;                                   // 1: point stride 1 on all three tensors (channel-planar rows): load_records' coalesced path
    int elem_size;                             // sizeof(T) of the correspondence tensors
    int nca, nla;                              // LDS carve: chunk slots (multiple of 4), pairwise leaves (>= 1)
    const void *K; int K_stride; int K_f64;    // K_stride 0 (broadcast) or 9
    const void *ur, *vr; int r_stride; int r_f64;
    const float *ransac_thr;
    const double *init_pose;
    int B, P;
    double z_min; float istd_thres; int inlier_opt_only; int flags; int mean_mode;
    int lm_max_iter;                           // Ceres max_num_iterations (50 unless MR_LM_MAXIT bits are set)
    uint8_t *valid; float *pose; float *cov; float *tr; uint8_t *mask; float *diag;
    double *pose64, *cov64, *tr64;            // legacy per-object ABI outputs (nullable)
    unsigned long long *stamps;               // debug: (B,24) s_memtime stamps (nullable)
    const float *calib_logscale; float corr_sd; float *cov_calib;   // optional fused R8/R13 epilogue: calibrated + distance-corrected covariance
    int from_head;                            // 1: the tile is decoded in-kernel from the raw NOC-head output (`dec`)
    DecodeArgs dec;
    PairwisePlan plan;
    const uint8_t *init_mask, *init_valid;    // EXT launches only (appended: the offsets of everything above are what the tuned kernels read)
};
// EXT launches over the objects of several calls (mr_pnp_uncert_from_init_grouped; a SECOND kernel argument of a kernel of its own, so that
// the argument block — and with it the code — of every other instantiation stays what it was): object b belongs to call b / group_B, whose
// pointers — biased on the host so that the GLOBAL object index addresses them — replace PnpArgs'
struct PnpCallTable {
    int ncalls, group_B;
    struct CallPtrs {
        const void *x2d, *istd, *x3d, *K, *ur, *vr;
        const double *init_pose; const uint8_t *init_mask, *init_valid;
        uint8_t *valid; float *pose, *cov, *tr; uint8_t *mask; float *diag; float *cov_calib;
    } call[8];          // = kEpMaxGroup
};

#include "pnp_kernel.inc"
#include "pnp6_kernel.inc"
#include "hessian_kernel.inc"
#include "pnp_noc_kernel.inc"
#include "epnp_kernel.inc"
#include "epnp_eig_low4.inc"
#include "epnp_stages.inc"
constexpr size_t kNocLds = sizeof(double) * (2 * 4 * kRedN + 2 * 40);     // reduction scratch + two sets of block sums

// ------------------------------------------------------------------------------------------------
// N1: rotated-BEV NMS, the consumer that follows the PnP (monorun_roi_head.py:619-655 calls
// mmdet3d.ops.iou3d.nms_gpu — third-party, not in the reference tree; restated from its published algorithm:
// sort by score, rotated-rectangle IoU = overlap / max(area_a + area_b - overlap, 1e-8), greedy
// suppression of IoU > thr).  One workgroup per class group (n <= kNmsMax boxes).
constexpr int kNmsMax = 512;

struct NmsBox { float cx, cy; float px[4], py[4]; float area; };   // CCW corners relative to nothing (absolute)

// length-weighted boundary integral of the part of segment p->p+d that lies inside the convex CCW polygon q
// (Cyrus-Beck parametric clipping, no dynamic arrays); CLOSED selects >= (boundary counts) or > (it does not),
// so that an edge shared by both rectangles is counted exactly once.
template <bool CLOSED>
__device__ __forceinline__ float edge_inside_area(float px, float py, float dx, float dy, const float (&qx)[4], const float (&qy)[4]) {
    float t0 = 0.0f, t1 = 1.0f;
    bool empty = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ax = qx[e], ay = qy[e], bx = qx[(e + 1) & 3], by = qy[(e + 1) & 3];
        const float nx = -(by - ay), ny = bx - ax;                 // inward normal of a CCW edge
        const float f0 = nx * (px - ax) + ny * (py - ay);
        const float den = nx * dx + ny * dy;
        if (den > 0.0f) t0 = fmaxf(t0, -f0 / den);
        else if (den < 0.0f) t1 = fminf(t1, -f0 / den);
        else if (CLOSED ? (f0 < 0.0f) : (f0 <= 0.0f)) empty = true;
    }
    if (empty || !(t1 > t0)) return 0.0f;
    const float x0 = px + t0 * dx, y0 = py + t0 * dy, x1 = px + t1 * dx, y1 = py + t1 * dy;
    return 0.5f * (x0 * y1 - x1 * y0);
}

__device__ __forceinline__ float rotated_iou(const NmsBox &a, const NmsBox &b) {
    // work relative to a's centre to keep fp32 cancellation small
    float ax[4], ay[4], bx[4], by[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ax[i] = a.px[i] - a.cx; ay[i] = a.py[i] - a.cy; bx[i] = b.px[i] - a.cx; by[i] = b.py[i] - a.cy; }
    float ov = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ov += edge_inside_area<true>(ax[i], ay[i], ax[(i + 1) & 3] - ax[i], ay[(i + 1) & 3] - ay[i], bx, by);
        ov += edge_inside_area<false>(bx[i], by[i], bx[(i + 1) & 3] - bx[i], by[(i + 1) & 3] - by[i], ax, ay);
    }
    ov = fmaxf(ov, 0.0f);
    return ov / fmaxf(a.area + b.area - ov, 1e-8f);
}

__global__ void __launch_bounds__(256) nms_bev_kernel(const float *boxes, const float *scores, const int *offsets, float thr,
                                                      long long *keep, int *num_keep) {
    const int g = blockIdx.x, tid = threadIdx.x;
    const int off = offsets[g], n = offsets[g + 1] - off;
    extern __shared__ __align__(16) unsigned char smem[];
    int np2 = 1; while (np2 < n) np2 <<= 1;
    float *skey = (float *)smem;                         // [np2]
    int *sidx = (int *)(skey + np2);                     // [np2]
    NmsBox *sbox = (NmsBox *)(sidx + np2);               // [n] in sorted order
    const int nw = (n + 31) >> 5;
    unsigned *srow = (unsigned *)(sbox + n);             // [n][nw] suppression bits (j > i, IoU > thr)
    if (n <= 0) { if (tid == 0) num_keep[g] = 0; return; }
    for (int i = tid; i < np2; i += 256) { skey[i] = (i < n) ? scores[off + i] : -__int_as_float(0x7f800000); sidx[i] = (i < n) ? i : 0x7fffffff; }
    __syncthreads();
    // bitonic sort: descending score, ties by ascending index; NaN scores sort last
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const float ki = skey[i], kl = skey[l]; const int ii = sidx[i], il = sidx[l];
                    // "i before l" in the final order
                    const bool i_first = (ki > kl) || (ki == kl && ii < il) || (kl != kl && ki == ki);
                    const bool up = (i & k) == 0;
                    if (up ? !i_first : i_first) { skey[i] = kl; skey[l] = ki; sidx[i] = il; sidx[l] = ii; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 256) {
        const float *b = boxes + (long long)(off + sidx[i]) * 5;
        const float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3], ang = b[4];
        NmsBox nb;
        nb.cx = 0.5f * (x1 + x2); nb.cy = 0.5f * (y1 + y2);
        const float hw = 0.5f * (x2 - x1), hh = 0.5f * (y2 - y1);
        float sn, cs; sincosf(ang, &sn, &cs);
        const float ddx[4] = { -hw, hw, hw, -hw }, ddy[4] = { -hh, -hh, hh, hh };
#pragma unroll
        for (int c = 0; c < 4; ++c) { nb.px[c] = nb.cx + ddx[c] * cs + ddy[c] * sn; nb.py[c] = nb.cy - ddx[c] * sn + ddy[c] * cs; }
        if (hw * hh < 0.0f) {                               // keep the corner order counter-clockwise
            const float tx = nb.px[1], ty = nb.py[1]; nb.px[1] = nb.px[3]; nb.py[1] = nb.py[3]; nb.px[3] = tx; nb.py[3] = ty;
        }
        nb.area = fabsf((x2 - x1) * (y2 - y1));
        sbox[i] = nb;
    }
    __syncthreads();
    for (int t = tid; t < n * nw; t += 256) {
        const int i = t / nw, w = t - i * nw;
        unsigned bits = 0;
        const NmsBox a = sbox[i];
        for (int jj = 0; jj < 32; ++jj) {
            const int j = w * 32 + jj;
            if (j > i && j < n && rotated_iou(a, sbox[j]) > thr) bits |= 1u << jj;
        }
        srow[t] = bits;
    }
    __syncthreads();
    if (tid < 64) {                                          // one wave, wave-synchronous greedy pass
        unsigned removed = 0;                                // lane w holds word w of the removed set (nw <= 16)
        int kept = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned word = __builtin_amdgcn_readlane(removed, i >> 5);
            if (!((word >> (i & 31)) & 1u)) {
                if (tid == 0) keep[off + kept] = (long long)sidx[i];
                ++kept;
                if (tid < nw) removed |= srow[i * nw + tid];
            }
        }
        if (tid == 0) num_keep[g] = kept;
    }
}

__global__ void __launch_bounds__(64) spin_kernel(long long ticks) {
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// The reference's covariance fallback (pnp_uncert.py:77-85), per object: when torch.inverse raises, the reference keeps an
// object only if the smallest eigenvalue of its Hessian exceeds max(1e-6 * largest, 0), and sets the others to h := I.  The fused
// kernel reports "Cholesky failed" instead; this optional pass applies the eigenvalue rule to every object.  The eigenvalues of
// cov = h^-1 are the reciprocals of h's, so the rule reads lambda_min(cov) > max(1e-6 * lambda_max(cov), 0) on the matrix the
// kernel already wrote (cyclic Jacobi on the 4x4, fp64).  One thread per object.
__global__ void __launch_bounds__(64) cov_symeig_rule_kernel(uint8_t *valid, float *cov, int B, float *lam_out) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double A[16];
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 16; ++i) { A[i] = (double)cov[(long long)b * 16 + i]; finite = finite && isfinite(A[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) { const double m = 0.5 * (A[4 * i + j] + A[4 * j + i]); A[4 * i + j] = A[4 * j + i] = m; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dia += A[5 * i] * A[5 * i];
#pragma unroll
            for (int j = i + 1; j < 4; ++j) off += A[4 * i + j] * A[4 * i + j];
        }
        if (!(off > 1e-30 * dia)) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double apq = A[4 * p + q];
                if (apq != 0.0) {
                    const double theta = (A[5 * q] - A[5 * p]) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const double x = A[4 * k + p], y = A[4 * k + q]; A[4 * k + p] = c * x - sn * y; A[4 * k + q] = sn * x + c * y; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const double x = A[4 * p + k], y = A[4 * q + k]; A[4 * p + k] = c * x - sn * y; A[4 * q + k] = sn * x + c * y; }
                }
            }
    }
    const double lmin = fmin(fmin(A[0], A[5]), fmin(A[10], A[15])), lmax = fmax(fmax(A[0], A[5]), fmax(A[10], A[15]));
    if (lam_out) { lam_out[(long long)b * 2] = (float)lmin; lam_out[(long long)b * 2 + 1] = (float)lmax; }
    const bool keep = finite && (lmin > fmax(1e-6 * lmax, 0.0));
    if (!keep) {
        valid[b] = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) cov[(long long)b * 16 + i] = (i % 5 == 0) ? 1.0f : 0.0f;
    }
}

#include "kitti_eval_kernel.inc"

size_t lds_bytes(const PnpArgs &a, int wpo) {
    size_t n = 0;
    n += sizeof(double) * (2 * wpo * kRedN + kMsg);       // reduction scratch + leader/follower message + camera matrix
    n += sizeof(unsigned long long) * a.nca;
    n += sizeof(float) * kHyp * 8;
    n += sizeof(int) * wpo * kHyp;
    n += sizeof(float) * (4 * a.nla + 4);
    const bool small = wpo <= 2;                           // one- and two-wave instantiations: 12-byte B records at fp32, one index list (pnp_kernel.inc)
    n += (size_t)((small && a.elem_size == 4) ? 7 : 8) * a.P * a.elem_size;      // point records
    n += (small ? 1 : 2) * sizeof(uint16_t) * ((a.P + 7) & ~7);                   // candidate list (+ final inlier list)
    n += small ? sizeof(unsigned long long) * a.nca + 8 : (size_t)a.P;           // inlier mask: one bit (64-bit words on an 8-byte boundary: up to 4 bytes of padding) or one byte per point
    return (n + 15) & ~(size_t)15;
}

struct PlanNode { int left, right, height; };

int plan_rec(PairwisePlan &pl, std::vector<PlanNode> &nodes, int off, int n, bool &ok) {
    if (n <= 128) {                                     // numpy: n < 8 plain loop, n <= PW_BLOCKSIZE unrolled block
        if (pl.n_leaves >= kMaxLeaves) { ok = false; return 0; }
        pl.leaf_off[pl.n_leaves] = (uint16_t)off; pl.leaf_len[pl.n_leaves] = (uint16_t)n;
        return pl.n_leaves++;                           // slot of a leaf = its index
    }
    int n2 = n / 2; n2 -= n2 % 8;
    const int l = plan_rec(pl, nodes, off, n2, ok); if (!ok) return 0;
    const int r = plan_rec(pl, nodes, off + n2, n - n2, ok); if (!ok) return 0;
    auto height = [&](int s) { return s < 0 ? nodes[-s - 1].height : 0; };
    nodes.push_back({ l, r, 1 + (height(l) > height(r) ? height(l) : height(r)) });
    return -(int)nodes.size();                          // internal nodes: negative ids until renumbered
}

bool build_plan(PairwisePlan &pl, int P) {
    memset(&pl, 0, sizeof pl);
    std::vector<PlanNode> nodes;
    bool ok = true;
    const int root = plan_rec(pl, nodes, 0, P, ok);
    if (!ok || nodes.size() > (size_t)kMaxLeaves) return false;
    // order internal nodes by height (stable), renumber
    std::vector<int> order(nodes.size()), newid(nodes.size());
    int maxh = 0;
    for (auto &nd : nodes) if (nd.height > maxh) maxh = nd.height;
    if (maxh + 1 >= 16) return false;
    int k = 0;
    for (int h = 1; h <= maxh; ++h) {
        pl.level_start[h - 1] = (uint8_t)k;
        for (size_t i = 0; i < nodes.size(); ++i) if (nodes[i].height == h) { order[k] = (int)i; newid[i] = k; ++k; }
    }
    pl.level_start[maxh] = (uint8_t)k;
    pl.n_levels = maxh; pl.n_internal = (int)nodes.size();
    auto slot = [&](int s) { return s >= 0 ? s : pl.n_leaves + newid[-s - 1]; };
    for (int i = 0; i < pl.n_internal; ++i) { pl.left[i] = (uint8_t)slot(nodes[order[i]].left); pl.right[i] = (uint8_t)slot(nodes[order[i]].right); }
    pl.root = slot(root);
    return true;
}

std::atomic<int> g_last_hip_error{0};
unsigned long long *g_stamps = nullptr;
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_last_hip_error = (int)e_; return MR_ERR_HIP; } } while (0)

// What the heuristics below need to know about the device, read once per device from hipGetDeviceProperties (an MI355X reports
// 256 CUs and 160 KB of LDS per CU; a partitioned or future part reports its own).  CDNA compute units have 4 SIMDs.
struct DevInfo { int cus; size_t lds_per_cu; };
DevInfo dev_info() {
    static std::mutex mu; static DevInfo cache[kMaxDevices]; static bool have[kMaxDevices] = {};
    int dev = 0;
    DevInfo d = { 256, (size_t)160 * 1024 };
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return d;
    std::lock_guard<std::mutex> lk(mu);
    if (!have[dev]) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) == hipSuccess) {
            if (pr.multiProcessorCount > 0) d.cus = pr.multiProcessorCount;
            if (pr.maxSharedMemoryPerMultiProcessor > 0) d.lds_per_cu = pr.maxSharedMemoryPerMultiProcessor;
        }
        cache[dev] = d; have[dev] = true;
    }
    return cache[dev];
}
constexpr int kSimdsPerCu = 4;

template <typename T, int WPO>
int launch(const PnpArgs &a, hipStream_t st) {
    const size_t lds = lds_bytes(a, WPO);
    if (lds > dev_info().lds_per_cu) return MR_ERR_UNSUPPORTED;
    if (lds > 48 * 1024) {
        // the opt-in is a per-device function attribute: remember what was granted on each device
        static std::mutex mu; static size_t granted[kMaxDevices] = {};
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= kMaxDevices || lds > granted[dev]) {
            HIP_TRY(hipFuncSetAttribute((const void *)pnp_uncert_kernel<T, WPO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < kMaxDevices) granted[dev] = lds;
        }
    }
    if (a.flags & MR_ANY_ORDER)      // no barrier bit on the dispatch packet: the launch need not wait for earlier launches on this stream
        hipExtLaunchKernelGGL((pnp_uncert_kernel<T, WPO>), dim3(a.B), dim3(64 * WPO), (uint32_t)lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, a);
    else
        hipLaunchKernelGGL((pnp_uncert_kernel<T, WPO>), dim3(a.B), dim3(64 * WPO), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

int grant_lds(const void *fn, size_t lds);

template <typename T, int WPO>
int launch_ext(const PnpArgs &a, hipStream_t st, const PnpCallTable *tbl = nullptr, const EpnpRefitIn *rf = nullptr) {
    const size_t lds = lds_bytes(a, WPO);
    if (lds > dev_info().lds_per_cu) return MR_ERR_UNSUPPORTED;
    if (rf) {                               // the initialiser's re-fit as this launch's prologue (mr_pnp_uncert_from_epnp_grouped)
        if (!tbl) return MR_ERR_BAD_ARGUMENT;
        int r;
        if (lds > 48 * 1024 && (r = grant_lds((const void *)pnp_uncert_refit_kernel<T, WPO>, lds)) != MR_OK) return r;
        hipLaunchKernelGGL((pnp_uncert_refit_kernel<T, WPO>), dim3(a.B), dim3(64 * WPO), lds, st, a, *tbl, *rf);
        HIP_TRY(hipGetLastError());
        return MR_OK;
    }
    if (tbl) {                              // the objects of several calls: the kernel that takes the call table as a second argument
        if (lds > 48 * 1024) {
            static std::mutex mu; static size_t granted[kMaxDevices] = {};
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            std::lock_guard<std::mutex> lk(mu);
            if (dev < 0 || dev >= kMaxDevices || lds > granted[dev]) {
                HIP_TRY(hipFuncSetAttribute((const void *)pnp_uncert_group_kernel<T, WPO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                if (dev >= 0 && dev < kMaxDevices) granted[dev] = lds;
            }
        }
        hipLaunchKernelGGL((pnp_uncert_group_kernel<T, WPO>), dim3(a.B), dim3(64 * WPO), lds, st, a, *tbl);
        HIP_TRY(hipGetLastError());
        return MR_OK;
    }
    if (lds > 48 * 1024) {
        static std::mutex mu; static size_t granted[kMaxDevices] = {};
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= kMaxDevices || lds > granted[dev]) {
            HIP_TRY(hipFuncSetAttribute((const void *)pnp_uncert_kernel<T, WPO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < kMaxDevices) granted[dev] = lds;
        }
    }
    if (a.flags & MR_ANY_ORDER)      // no barrier bit on the dispatch packet: the launch starts once the launch in front of it has STARTED (the LM launches
        // of the calls of one launch set: the first waits for the set's initialiser launches, the others run beside it — PnPEpnpGroupLaunch)
        hipExtLaunchKernelGGL((pnp_uncert_kernel<T, WPO, true>), dim3(a.B), dim3(64 * WPO), (uint32_t)lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, a);
    else
        hipLaunchKernelGGL((pnp_uncert_kernel<T, WPO, true>), dim3(a.B), dim3(64 * WPO), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

template <typename T>
int launch_wpo(PnpArgs &a, int wpo, hipStream_t st, const PnpCallTable *tbl = nullptr, const EpnpRefitIn *rf = nullptr) {
    a.elem_size = (int)sizeof(T);
    a.vec = (!a.from_head && a.s2[1] == 1 && a.sw[1] == 1 && a.s3[1] == 1) ? 1 : 0;      // channel-planar rows: coalesced per-point loads
    a.nca = (((a.P + 63) / 64) + 3) & ~3;
    a.nla = a.plan.n_leaves > 0 ? a.plan.n_leaves : 1;
    { const int mi = (a.flags & MR_LM_MAXIT_MASK) >> MR_LM_MAXIT_SHIFT; a.lm_max_iter = mi ? mi : 50; }
    if (a.init_mask) {                      // external initialiser: 2, 4 or 8 waves per object
        if (wpo < 2) wpo = 2;
        if (wpo == 3) wpo = 4;
        switch (wpo) {
            case 2: return launch_ext<T, 2>(a, st, tbl, rf);
            case 4: return launch_ext<T, 4>(a, st, tbl, rf);
            case 8: return launch_ext<T, 8>(a, st, tbl, rf);
            default: return MR_ERR_BAD_ARGUMENT;
        }
    }
    if (tbl) return MR_ERR_BAD_ARGUMENT;
    switch (wpo) {
        case 1: return launch<T, 1>(a, st);
        case 2: return launch<T, 2>(a, st);
        case 3: return launch<T, 3>(a, st);
        case 4: return launch<T, 4>(a, st);
        case 8: return launch<T, 8>(a, st);
        default: return MR_ERR_BAD_ARGUMENT;
    }
}

int pick_wpo(int B, int P, int flags) {
    int w = (flags & MR_WAVES_MASK) >> MR_WAVES_SHIFT;
    if (w) return w;
    // fp32 variants hold 4 resident waves per SIMD (<= 128 VGPRs) -> 4096 waves on 256 CUs x 4 SIMDs.  More waves per
    // object shorten an object's latency chain (what bounds small batches), fewer waves cost fewer instructions per object
    // (what bounds large ones).  Measured on MI355X, P = 784: 4 waves/object wins up to B = 2048, 2 from B = 4096
    // (i.e. while B x waves x 2 does not exceed twice the resident-wave capacity of the chip).
    const long long wave_slots = (long long)dev_info().cus * kSimdsPerCu * 4;      // 4096 on an MI355X
    w = 1;
    while (w < 4 && (long long)B * w * 2 <= 2 * wave_slots && P >= 64 * w * 2) w *= 2;      // small batches: fill the SIMDs
    int wp = 1;
    while (wp < 4 && P > 64 * wp * 8) wp *= 2;                                     // large tiles: <= ~8 points per lane
    if (wp > w) w = wp;                                                            // (P = 784 -> 2, P = 3136 -> 4)
    return w;
}

// Tiles so large that at most two workgroups fit the LDS of a CU (config 5: 56x56 points, 66 KB as fp16, 100 KB as fp32 against 160 KB):
// with 4 waves per object a CU would hold 8 waves; 8 waves per object restore 16 (4 per SIMD — the 128-VGPR kernels allow it).
// Measured on the config-5 shard (8192 objects, fp16): 0.851 -> 0.810 ms.
int widen_for_large_tiles(int wpo, const PnpArgs &a, int flags, int in_dtype) {
    if ((flags & MR_WAVES_MASK) || wpo != 4 || in_dtype == MR_F64 || a.P < 64 * 8 * 2) return wpo;
    PnpArgs t = a;
    t.elem_size = (in_dtype == MR_F16) ? 2 : 4;                 // the launcher sets it later, from the template type
    return (lds_bytes(t, 4) * 3 > dev_info().lds_per_cu) ? 8 : wpo;
}

// 6-DoF refinement (second launch of pnp_uncert(..., use_6dof=True)): see pnp6_kernel.inc
template <typename T>
int launch_pnp6(Pnp6Args &a, hipStream_t st) {
    a.elem_size = (int)sizeof(T);
    a.vec = (a.s2[1] == 1 && a.sw[1] == 1 && a.s3[1] == 1) ? 1 : 0;
    const int nchunk = (a.P + 63) / 64;
    const size_t lds = sizeof(double) * 2 * 4 * kRedN + sizeof(unsigned long long) * ((nchunk + 3) & ~3) + (size_t)8 * a.P * sizeof(T) +
                       sizeof(uint16_t) * ((a.P + 7) & ~7) + 16;
    if (lds > dev_info().lds_per_cu) return MR_ERR_UNSUPPORTED;
    if (lds > 48 * 1024) {
        // per-device function attribute, raised monotonically under a lock (two host threads with different P must not shrink it
        // between the other's set and launch), and no driver call in the steady state — the pattern of launch<T, WPO>
        static std::mutex mu; static size_t granted[kMaxDevices] = {};
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= kMaxDevices || lds > granted[dev]) {
            HIP_TRY(hipFuncSetAttribute((const void *)pnp6_refine_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < kMaxDevices) granted[dev] = lds;
        }
    }
    hipLaunchKernelGGL((pnp6_refine_kernel<T>), dim3(a.B), dim3(kThreads6), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

// opt-in to more than 64 KB of dynamic LDS, once per (kernel, device, size)
int grant_lds(const void *fn, size_t lds) {
    static std::mutex mu; static std::map<std::pair<const void *, int>, size_t> granted;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    size_t &g = granted[std::make_pair(fn, dev)];
    if (lds > g) {
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        g = lds;
    }
    return MR_OK;
}

// The staged form of the initialiser (epnp_stages.inc): six or seven launches on `st` (the second round idles when no object needs it: one launch for small sets, two beyond), intermediate results in `workspace` (caller's, at
// least mr_epnp_workspace_bytes(B, P)) or, when that is null, in a stream-ordered allocation of the device's default memory pool.
template <typename T>
int launch_epnp_stages(EpnpStageArgs &ea, void *workspace, size_t workspace_bytes, int first_round, hipStream_t st) {
    PnpArgs &a = ea.p;
    a.elem_size = (int)sizeof(T);
    a.vec = (a.s2[1] == 1 && a.sw[1] == 1 && a.s3[1] == 1) ? 1 : 0;
    a.nca = (((a.P + 63) / 64) + 3) & ~3;
    a.nla = a.plan.n_leaves > 0 ? a.plan.n_leaves : 1;
#ifndef MR_EP_LDS_PAD_CONS
#define MR_EP_LDS_PAD_CONS 0      // development aid: extra LDS per workgroup of the consensus / re-fit launch (residency experiments)
#endif
#ifndef MR_EP_LDS_PAD_REFIT
#define MR_EP_LDS_PAD_REFIT 0
#endif
    const size_t lds_f = epnp_front_lds_bytes(a), lds_c = epnp_consensus_lds_bytes(a) + MR_EP_LDS_PAD_CONS, lds_r = epnp_refit_lds_bytes(a) + MR_EP_LDS_PAD_REFIT;
    if (lds_f > dev_info().lds_per_cu || lds_c > dev_info().lds_per_cu || lds_r > dev_info().lds_per_cu) return MR_ERR_UNSUPPORTED;
    const size_t need = epnp_work_bytes(a.B, a.P, nullptr, nullptr);
    unsigned char *base = (unsigned char *)workspace;
    bool own = false;
    if (base) { if (workspace_bytes < need || ((uintptr_t)base & 255)) return MR_ERR_BAD_ARGUMENT; }
    else {
        // a PRIVATE stream-ordered pool per device (the process's default pool is left as it is): freed workspaces stay in it across
        // synchronisations (release threshold = max), so the steady state allocates nothing
        static std::mutex mu; static hipMemPool_t pools[kMaxDevices] = {};
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= kMaxDevices) return MR_ERR_UNSUPPORTED;
        hipMemPool_t pool;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!pools[dev]) {
                hipMemPoolProps props;
                memset(&props, 0, sizeof props);
                props.allocType = hipMemAllocationTypePinned;
                props.handleTypes = hipMemHandleTypeNone;
                props.location.type = hipMemLocationTypeDevice;
                props.location.id = dev;
                HIP_TRY(hipMemPoolCreate(&pools[dev], &props));
                uint64_t keep = ~0ull;
                HIP_TRY(hipMemPoolSetAttribute(pools[dev], hipMemPoolAttrReleaseThreshold, &keep));
            }
            pool = pools[dev];
        }
        HIP_TRY(hipMallocFromPoolAsync((void **)&base, need, pool, st));
        own = true;
    }
    epnp_work_bytes(a.B, a.P, &ea.w, base);
    int rc = MR_OK;
    auto run = [&]() -> int {
        int r;
        if ((r = grant_lds((const void *)epnp_front_kernel<T>, lds_f)) != MR_OK) return r;
        // waves per object of the consensus launch: 4; the environment variable MR_EP_CONS_WPO=2 selects the two-wave instantiation (same bits; measured
        // slower one call at a time AND in launch sets — 11.39 -> 11.21 M solves/s with sets of five, 11.93 -> 11.51 with sets of eight: the launch's
        // time is throughput work, not waves waiting — profiles/r05_epnp_grouping.txt)
        static const int cons_env = [] { const char *e = getenv("MR_EP_CONS_WPO"); const int v = e ? atoi(e) : 0; return (v == 2 || v == 4) ? v : 0; }();
        const int cons_wpo = cons_env ? cons_env : 4;
        if ((r = grant_lds(cons_wpo == 2 ? (const void *)epnp_consensus_kernel<T, 2> : (const void *)epnp_consensus_kernel<T, 4>, lds_c)) != MR_OK) return r;
        if ((r = grant_lds((const void *)epnp_refit_kernel<T>, lds_r)) != MR_OK) return r;
        hipLaunchKernelGGL((epnp_front_kernel<T>), dim3(a.B), dim3(kEpThreads), lds_f, st, ea);
        // The 30 hypotheses of an object are solved in two rounds: [0, first) for every object, the rest only for the objects whose
        // replayed loop still wants iterations after `first` (ptsetreg.cpp's adaptive bound: with few outliers it drops to a
        // handful after the first good model — config-2 batches: 1.5 iterations on average, 8 at most).  Same results either way.
        const int first = first_round < 1 ? 1 : (first_round > kEpMaxIters ? kEpMaxIters : first_round);
        // small launch sets (one call at a time): the second round as ONE launch (epnp_round2_kernel); launch sets in flight keep the two compact ones
        // development / tests: force the quads per matrix (0, 2 or 4 levels) of both lane-mapped launches (MR_EP_WIDE) or of one (MR_EP_WIDE_HYP, MR_EP_WIDE_BETAS)
        static const auto lv_env = [](const char *name) { const char *e = getenv(name); const int v = e ? atoi(e) : -1; return (v == 0 || v == 2 || v == 4) ? v : -1; };
        static const int wide_env = lv_env("MR_EP_WIDE");
        static const int wide_hyp_env = lv_env("MR_EP_WIDE_HYP") >= 0 ? lv_env("MR_EP_WIDE_HYP") : wide_env, wide_betas_env = lv_env("MR_EP_WIDE_BETAS") >= 0 ? lv_env("MR_EP_WIDE_BETAS") : wide_env;
        static const int r2_env = [] { const char *e = getenv("MR_EP_ROUND2"); return e ? atoi(e) : 0; }();      // development: 1 = always two launches, 2 = always one
        const bool one_launch_round2 = cons_wpo == 4 && (kEpMaxIters - first) <= kEpRound2Quads && first < kEpMaxIters && (r2_env == 2 || (r2_env == 0 && a.B < 2048));
        for (int round = 0; round < 2; ++round) {
            ea.h0 = round == 0 ? 0 : first; ea.h1 = round == 0 ? first : kEpMaxIters;
            const int nh = ea.h1 - ea.h0;
            if (nh <= 0) break;
            if (round == 1 && one_launch_round2) {
                const size_t lds_2 = epnp_round2_lds_bytes(a) + MR_EP_LDS_PAD_CONS;
                if ((r = grant_lds((const void *)epnp_round2_kernel<T>, lds_2)) != MR_OK) return r;
                hipLaunchKernelGGL((epnp_round2_kernel<T>), dim3(a.B), dim3(256), lds_2, st, ea);
                break;
            }
            const long long quads = (long long)a.B * nh;
            // 16 quads per single-wave workgroup: 8 / 4 per wave (more waves, fewer matrices in lockstep) measured 74 / 140 us against 74 us one call
            // at a time and 5.4 / 4.1 against 6.3 M solves/s in flight (profiles/r04_epnp_quads_per_wave.txt)
            // wide form (a wave per hypothesis) while that still leaves SIMDs without a wave: up to 1024 hypotheses (one image's <= 100 proposals x the first
            // round of 10).  Measured, one call at a time (profiles/r06_wide_sweep.txt): B = 100: -6 us; a 16-lane row per hypothesis at B = 200 / 320: +-0; a wave
            // at B = 200: +40 us (2000 waves: the chip is full and its clock drops)
            const int lv_h = wide_hyp_env >= 0 ? wide_hyp_env : (quads <= 1024 ? 4 : (quads <= 4096 ? 2 : 0));      // (rows up to 4096 hypotheses: B = 128 ... 400: -9 ... -3 us; 5120: +30)
            if (lv_h == 4) hipLaunchKernelGGL(epnp_hyp_kernel<4>, dim3((unsigned)quads), dim3(64), 0, st, ea);
            else if (lv_h == 2) hipLaunchKernelGGL(epnp_hyp_kernel<2>, dim3((unsigned)((quads + 3) / 4)), dim3(64), 0, st, ea);
            else hipLaunchKernelGGL(epnp_hyp_kernel<0>, dim3((unsigned)((quads + 15) / 16)), dim3(64), 0, st, ea);
            if (cons_wpo == 2) hipLaunchKernelGGL((epnp_consensus_kernel<T, 2>), dim3(a.B), dim3(128), lds_c, st, ea);
            else hipLaunchKernelGGL((epnp_consensus_kernel<T, 4>), dim3(a.B), dim3(256), lds_c, st, ea);
        }
        {   // (quad form with 8 / 4 / 2 quads per wave: 67 / 68 / 102 us against 55 us, round 4)
            // a wave per object up to 512 objects, a 16-lane row up to 2047 (B = 100: -10 us, 512: -5, 1024: -5 with rows, +16 with waves), the quad form for launch sets
            const int lv_b = wide_betas_env >= 0 ? wide_betas_env : (a.B <= 512 ? 4 : (a.B < 2048 ? 2 : 0));
            if (lv_b == 4) hipLaunchKernelGGL(epnp_refit_betas_kernel<4>, dim3((unsigned)a.B), dim3(64), 0, st, ea);
            else if (lv_b == 2) hipLaunchKernelGGL(epnp_refit_betas_kernel<2>, dim3((unsigned)((a.B + 3) / 4)), dim3(64), 0, st, ea);
            else hipLaunchKernelGGL(epnp_refit_betas_kernel<0>, dim3((unsigned)((a.B + 15) / 16)), dim3(64), 0, st, ea);
        }
        if (!(a.flags & MR_EPNP_DEFER_REFIT))                  // else: the LM launch carries it (mr_pnp_uncert_from_epnp_grouped)
            hipLaunchKernelGGL((epnp_refit_kernel<T>), dim3(a.B), dim3(kEpPoseThreads), lds_r, st, ea);
        HIP_TRY(hipGetLastError());
        return MR_OK;
    };
    rc = run();
    if (own) { const hipError_t e = hipFreeAsync(base, st); if (rc == MR_OK && e != hipSuccess) { g_last_hip_error = (int)e; rc = MR_ERR_HIP; } }
    return rc;
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

// development aid (not in the public header): device buffer of (B,10) u64 cycle stamps, or NULL to disable
void mr_pnp_debug_set_stamps(unsigned long long *dev_ptr) { g_stamps = dev_ptr; }

// Occupies one wavefront of the device for `microseconds` (100 MHz constant clock).  PnPPipeline uses it to find out which of
// its streams the runtime really runs side by side: HIP maps streams onto a small number of hardware queues (4 per priority level
// by default) and two streams that share a queue serialise.
int mr_spin(int microseconds, void *stream) {
    if (microseconds < 0 || microseconds > 1000000) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

// waves per object the library would pick for a launch of `objects_in_flight` objects x P points on the current device (pick_wpo): lets
// a caller that keeps several launches in flight apply the library's own rule to ALL the objects on the chip (PnPPipeline.flags_for)
int mr_pick_waves(int objects_in_flight, int P) {
    if (objects_in_flight < 1 || P < 4) return MR_ERR_BAD_ARGUMENT;
    return pick_wpo(objects_in_flight, P, 0);
}

int mr_pnp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int pnp_uncert_launch(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    const float *ransac_thr, const double *init_pose, const uint8_t *init_mask, const uint8_t *init_valid, int B, int P,
    float z_min, float istd_thres, int inlier_opt_only, int flags,
    uint8_t *valid, float *pose, float *cov, float *tr_radius, uint8_t *inlier_mask, float *diag, void *stream,
    int ncalls = 1, const PnpCallTable::CallPtrs *calls = nullptr, const EpnpRefitIn *rf = nullptr, const float *calib_logscale = nullptr, float corr_sd = 0.0f) {
    if (B < 0 || P < 4 || P > 64 * kMaxChunks) return MR_ERR_BAD_ARGUMENT;
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
    a.ransac_thr = ransac_thr; a.init_pose = init_pose; a.init_mask = init_mask; a.init_valid = init_valid;
    a.B = B; a.P = P; a.z_min = (double)z_min; a.istd_thres = istd_thres; a.inlier_opt_only = inlier_opt_only; a.flags = flags;
    a.valid = valid; a.pose = pose; a.cov = cov; a.tr = tr_radius; a.mask = inlier_mask; a.diag = diag;
    a.stamps = g_stamps;
    if (calib_logscale && calls && calls[0].cov_calib) { a.calib_logscale = calib_logscale; a.corr_sd = corr_sd; a.cov_calib = calls[0].cov_calib; }      // per call: the table's
    int mm = flags & MR_MEAN_MASK;
    if (mm == MR_MEAN_AUTO) mm = (istd_strides[1] == 1 && P > 1) ? MR_MEAN_PAIRWISE : MR_MEAN_SEQUENTIAL;
    a.mean_mode = mm;
    if (mm == MR_MEAN_PAIRWISE && !(flags & MR_NO_ISTD_MASK) && !init_mask) {
        if (!build_plan(a.plan, P)) return MR_ERR_UNSUPPORTED;
    }
    PnpCallTable tbl;
    memset(&tbl, 0, sizeof tbl);
    tbl.ncalls = 1; tbl.group_B = B;
    if (ncalls > 1 || rf) {                             // a launch over the objects of several calls (EXT only): mr_pnp_uncert_from_init_grouped / _from_epnp_grouped
        tbl.ncalls = ncalls; a.B = B * ncalls;
        for (int c = 0; c < ncalls; ++c) tbl.call[c] = calls[c];
    }
    const int wpo = widen_for_large_tiles(pick_wpo(a.B, P, flags), a, flags, in_dtype);
    hipStream_t st = (hipStream_t)stream;
    const PnpCallTable *tp = (ncalls > 1 || rf) ? &tbl : nullptr;
    switch (in_dtype) {
        case MR_F32: return launch_wpo<float>(a, wpo, st, tp, rf);
        case MR_F16: return launch_wpo<__half>(a, wpo, st, tp, rf);
        case MR_F64: return launch_wpo<double>(a, wpo, st, tp, rf);
        default: return MR_ERR_UNSUPPORTED;
    }
}

int mr_pnp_uncert_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    const float *ransac_thr, const double *init_pose, int B, int P,
    float z_min, float istd_thres, int inlier_opt_only, int flags,
    uint8_t *valid, float *pose, float *cov, float *tr_radius, uint8_t *inlier_mask, float *diag, void *stream) {
    return pnp_uncert_launch(x2d, x2d_strides, istd, istd_strides, x3d, x3d_strides, in_dtype, cam_mats, cam_batch, u_range, v_range, range_batch,
                             ransac_thr, init_pose, nullptr, nullptr, B, P, z_min, istd_thres, inlier_opt_only, flags,
                             valid, pose, cov, tr_radius, inlier_mask, diag, stream);
}

int mr_pnp_uncert_from_init_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    const double *init_pose, const uint8_t *init_mask, const uint8_t *init_valid, int B, int P,
    float z_min, int inlier_opt_only, int flags,
    uint8_t *valid, float *pose, float *cov, float *tr_radius, uint8_t *inlier_mask, float *diag, void *stream) {
    if (B > 0 && (!init_pose || !init_mask || !init_valid)) return MR_ERR_BAD_ARGUMENT;
    return pnp_uncert_launch(x2d, x2d_strides, istd, istd_strides, x3d, x3d_strides, in_dtype, cam_mats, cam_batch, u_range, v_range, range_batch,
                             nullptr, init_pose, init_mask, init_valid, B, P, z_min, 0.0f, inlier_opt_only, flags,
                             valid, pose, cov, tr_radius, inlier_mask, diag, stream);
}

static int pnp_from_init_grouped(
    int ncalls, const void *const *x2d, const int64_t *x2d_strides, const void *const *istd, const int64_t *istd_strides,
    const void *const *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *const *cam_mats, int cam_batch, const float *const *u_range, const float *const *v_range, int range_batch,
    const double *const *init_pose, const uint8_t *const *init_mask, const uint8_t *const *init_valid, int B, int P,
    float z_min, int inlier_opt_only, int flags,
    uint8_t *const *valid, float *const *pose, float *const *cov, float *const *tr_radius, uint8_t *const *inlier_mask, float *const *diag, void *stream,
    EpnpRefitIn *rf = nullptr, float *const *epnp_diag = nullptr, const float *calib_logscale = nullptr, float corr_sd = 0.0f, float *const *cov_calib = nullptr) {
    if (ncalls < 1 || ncalls > 8 || B < 0) return MR_ERR_BAD_ARGUMENT;
    const bool with_calib = cov_calib && cov_calib[0];
    if (with_calib && (!calib_logscale || (flags & MR_COV_NONE))) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if (!x2d || !istd || !x3d || !x2d_strides || !istd_strides || !x3d_strides || !cam_mats || !u_range || !v_range || !init_pose || !init_mask || !init_valid ||
        !valid || !pose || !tr_radius || (!cov && !(flags & MR_COV_NONE))) return MR_ERR_BAD_ARGUMENT;       // (no covariance asked: the table itself may be NULL, like its entries)
    if ((long long)B * ncalls > 0x7fffffffll / kEpMaxIters) return MR_ERR_UNSUPPORTED;                         // objects are numbered through the set in int arithmetic (as in epnp_ransac_launch)
    const bool with_mask = inlier_mask && inlier_mask[0], with_diag = diag && diag[0];
    const size_t esize = in_dtype == MR_F64 ? 8 : (in_dtype == MR_F32 ? 4 : 2);
    const long long ks = (cam_batch == 1) ? 0 : 9, rs = (range_batch == 1) ? 0 : 2;
    PnpCallTable::CallPtrs cp[8];
    for (int c = 0; c < ncalls; ++c) {
        if (!x2d[c] || !istd[c] || !x3d[c] || !cam_mats[c] || !u_range[c] || !v_range[c] || !init_pose[c] || !init_mask[c] || !init_valid[c] ||
            !valid[c] || !pose[c] || !tr_radius[c] || (!(cov && cov[c]) && !(flags & MR_COV_NONE))) return MR_ERR_BAD_ARGUMENT;
        if ((inlier_mask && inlier_mask[c] != nullptr) != with_mask || (diag && diag[c] != nullptr) != with_diag) return MR_ERR_BAD_ARGUMENT;      // all or none
        // pointers biased so that the GLOBAL object index c * B + i addresses object i of call c
        const long long o = (long long)c * B;
        PnpCallTable::CallPtrs &q = cp[c];
        q.x2d = (const char *)x2d[c] - o * x2d_strides[0] * (long long)esize;
        q.istd = (const char *)istd[c] - o * istd_strides[0] * (long long)esize;
        q.x3d = (const char *)x3d[c] - o * x3d_strides[0] * (long long)esize;
        q.K = (const char *)cam_mats[c] - o * ks * 4; q.ur = (const char *)u_range[c] - o * rs * 4; q.vr = (const char *)v_range[c] - o * rs * 4;
        q.init_pose = init_pose[c] - o * 4; q.init_mask = init_mask[c] - o * P; q.init_valid = init_valid[c] - o;
        q.valid = valid[c] - o; q.pose = pose[c] - o * 4; q.cov = (cov && cov[c]) ? cov[c] - o * 16 : nullptr; q.tr = tr_radius[c] - o;
        q.mask = with_mask ? inlier_mask[c] - o * P : nullptr; q.diag = with_diag ? diag[c] - o * 4 : nullptr;
        if ((cov_calib && cov_calib[c] != nullptr) != with_calib) return MR_ERR_BAD_ARGUMENT;
        q.cov_calib = with_calib ? cov_calib[c] - o * 16 : nullptr;
        if (rf) rf->diag[c] = (epnp_diag && epnp_diag[c]) ? epnp_diag[c] - o * 4 : nullptr;
    }
    return pnp_uncert_launch(x2d[0], x2d_strides, istd[0], istd_strides, x3d[0], x3d_strides, in_dtype, cam_mats[0], cam_batch, u_range[0], v_range[0], range_batch,
                             nullptr, init_pose[0], init_mask[0], init_valid[0], B, P, z_min, 0.0f, inlier_opt_only, flags,
                             valid[0], pose[0], cov ? cov[0] : nullptr, tr_radius[0], with_mask ? inlier_mask[0] : nullptr, with_diag ? diag[0] : nullptr, stream, ncalls, cp, rf, calib_logscale, corr_sd);
}

int mr_pnp_uncert_from_init_grouped(
    int ncalls, const void *const *x2d, const int64_t *x2d_strides, const void *const *istd, const int64_t *istd_strides,
    const void *const *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *const *cam_mats, int cam_batch, const float *const *u_range, const float *const *v_range, int range_batch,
    const double *const *init_pose, const uint8_t *const *init_mask, const uint8_t *const *init_valid, int B, int P,
    float z_min, int inlier_opt_only, int flags,
    uint8_t *const *valid, float *const *pose, float *const *cov, float *const *tr_radius, uint8_t *const *inlier_mask, float *const *diag, void *stream) {
    return pnp_from_init_grouped(ncalls, x2d, x2d_strides, istd, istd_strides, x3d, x3d_strides, in_dtype, cam_mats, cam_batch, u_range, v_range, range_batch,
                                 init_pose, init_mask, init_valid, B, P, z_min, inlier_opt_only, flags, valid, pose, cov, tr_radius, inlier_mask, diag, stream);
}

int mr_pnp_uncert_from_epnp_grouped(
    int ncalls, const void *const *x2d, const int64_t *x2d_strides, const void *const *istd, const int64_t *istd_strides,
    const void *const *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *const *cam_mats, int cam_batch, const float *const *u_range, const float *const *v_range, int range_batch,
    double *const *init_pose, const uint8_t *const *init_mask, uint8_t *const *init_valid, float *const *epnp_diag, int B, int P,
    float z_min, int inlier_opt_only, int flags,
    uint8_t *const *valid, float *const *pose, float *const *cov, float *const *tr_radius, uint8_t *const *inlier_mask, float *const *diag,
    const float *cov_calib_logscale, float cov_corr_sd, float *const *cov_calib,
    const void *workspace, size_t workspace_bytes, void *stream) {
    if (ncalls < 1 || ncalls > kEpMaxGroup || B < 0 || P < 4) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if ((long long)B * ncalls > 0x7fffffffll / kEpMaxIters) return MR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < epnp_work_bytes(B * ncalls, P, nullptr, nullptr)) return MR_ERR_BAD_ARGUMENT;
    EpnpRefitIn rf;
    memset(&rf, 0, sizeof rf);
    epnp_work_bytes(B * ncalls, P, &rf.w, (unsigned char *)const_cast<void *>(workspace));
    rf.B = (long long)B * ncalls;
    return pnp_from_init_grouped(ncalls, x2d, x2d_strides, istd, istd_strides, x3d, x3d_strides, in_dtype, cam_mats, cam_batch, u_range, v_range, range_batch,
                                 (const double *const *)init_pose, init_mask, (const uint8_t *const *)init_valid, B, P, z_min, inlier_opt_only, flags,
                                 valid, pose, cov, tr_radius, inlier_mask, diag, stream, &rf, epnp_diag, cov_calib_logscale, cov_corr_sd, cov_calib);
}

static int epnp_ransac_launch(
    int ncalls, const void *const *x2d, const int64_t *x2d_strides, const void *const *istd, const int64_t *istd_strides,
    const void *const *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *const *cam_mats, int cam_batch, const float *const *ransac_thr, int B, int P,
    float istd_thres, int flags, int max_iters,
    double *const *init_pose, uint8_t *const *init_mask, uint8_t *const *init_valid, float *const *diag, double *debug_hypotheses,
    void *workspace, size_t workspace_bytes, void *stream) {
    if (ncalls < 1 || ncalls > kEpMaxGroup || B < 0 || P < 4 || P > 64 * kMaxChunks || max_iters < 1 || max_iters > kEpMaxIters) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if ((long long)B * ncalls > 0x7fffffffll / kEpMaxIters) return MR_ERR_UNSUPPORTED;
    if (!x2d || !istd || !x3d || !x2d_strides || !istd_strides || !x3d_strides || !cam_mats || !init_pose || !init_mask || !init_valid) return MR_ERR_BAD_ARGUMENT;
    if (cam_batch != 1 && cam_batch != B) return MR_ERR_BAD_ARGUMENT;
    const bool with_thr = ransac_thr && ransac_thr[0], with_diag = diag && diag[0];
    for (int c = 0; c < ncalls; ++c) {
        if (!x2d[c] || !istd[c] || !x3d[c] || !cam_mats[c] || !init_pose[c] || !init_mask[c] || !init_valid[c]) return MR_ERR_BAD_ARGUMENT;
        if ((ransac_thr && ransac_thr[c] != nullptr) != with_thr || (diag && diag[c] != nullptr) != with_diag) return MR_ERR_BAD_ARGUMENT;     // all or none
    }
    if (debug_hypotheses && ncalls != 1) return MR_ERR_BAD_ARGUMENT;
    if ((flags & MR_EPNP_DEFER_REFIT) && !workspace) return MR_ERR_BAD_ARGUMENT;       // the LM launch that finishes the job needs the workspace
    const size_t esize = in_dtype == MR_F64 ? 8 : (in_dtype == MR_F32 ? 4 : 2);
    EpnpStageArgs sa;
    memset(&sa, 0, sizeof sa);
    PnpArgs &a = sa.p;
    a.x2d = x2d[0]; a.istd = istd[0]; a.x3d = x3d[0];
    for (int i = 0; i < 3; ++i) { a.s2[i] = x2d_strides[i]; a.sw[i] = istd_strides[i]; a.s3[i] = x3d_strides[i]; }
    a.K = cam_mats[0]; a.K_stride = (cam_batch == 1) ? 0 : 9; a.K_f64 = 0;
    a.ransac_thr = with_thr ? ransac_thr[0] : nullptr;
    a.B = B * ncalls; a.P = P; a.istd_thres = istd_thres; a.flags = flags;
    int mm = flags & MR_MEAN_MASK;
    if (mm == MR_MEAN_AUTO) mm = (istd_strides[1] == 1 && P > 1) ? MR_MEAN_PAIRWISE : MR_MEAN_SEQUENTIAL;
    a.mean_mode = mm;
    if (mm == MR_MEAN_PAIRWISE && !(flags & MR_NO_ISTD_MASK)) {
        if (!build_plan(a.plan, P)) return MR_ERR_UNSUPPORTED;
    }
    a.stamps = g_stamps;
    sa.init_pose = init_pose[0]; sa.init_mask = init_mask[0]; sa.init_ok = init_valid[0]; sa.diag = with_diag ? diag[0] : nullptr; sa.dbg_hyp = debug_hypotheses; sa.max_iters = max_iters;
    sa.ncalls = ncalls; sa.group_B = B;
    for (int c = 0; c < ncalls; ++c) {
        // pointers biased so that the GLOBAL object index c * B + i addresses object i of call c (epnp_stages.inc EpnpCallPtrs)
        const long long o = (long long)c * B;
        EpnpCallPtrs &q = sa.call[c];
        q.x2d = (const char *)x2d[c] - o * x2d_strides[0] * (long long)esize;
        q.istd = (const char *)istd[c] - o * istd_strides[0] * (long long)esize;
        q.x3d = (const char *)x3d[c] - o * x3d_strides[0] * (long long)esize;
        q.K = (const char *)cam_mats[c] - o * a.K_stride * (long long)sizeof(float);
        q.ransac_thr = with_thr ? ransac_thr[c] - o : nullptr;
        q.init_pose = init_pose[c] - o * 4; q.init_mask = init_mask[c] - o * P; q.init_ok = init_valid[c] - o; q.diag = with_diag ? diag[c] - o * 4 : nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    // hypotheses solved for every object before the replayed loop is consulted: MR_EPNP_FIRST_ROUND bits of `flags` (1..30), else the
    // environment variable MR_EPNP_FIRST_ROUND, else by the size of the launch set: 10 up to 2047 objects (one call at a time: the
    // second round is a full latency chain; 8 hypotheses make it idle in 85 % of config-2 batches, 10 in 97 %: 278.6 -> 264.1 us per
    // 1024-object call, 228.9 -> 221.8 at 256, within noise at 100 — profiles/r06_first_round.txt), 3 beyond (several calls grouped
    // or a large batch: the chip is busy, the hypotheses nobody needs are the cost — sets of three calls: 9.6 / 9.9 / 10.3 / 10.3 M solves/s
    // with 6 / 4 / 3 / 2, profiles/r05_epnp_grouped_first_round.txt).  Changes the work, never a result.
    static const int first_env = [] { const char *e = getenv("MR_EPNP_FIRST_ROUND"); const int v = e ? atoi(e) : 0; return v < 1 ? 0 : (v > 30 ? 30 : v); }();
    const int first_bits = (flags & MR_EPNP_FIRST_ROUND_MASK) >> MR_EPNP_FIRST_ROUND_SHIFT;
    const int first_round = first_bits ? (first_bits > 30 ? 30 : first_bits) : (first_env ? first_env : ((long long)B * ncalls >= 2048 ? 3 : 10));
    switch (in_dtype) {
        case MR_F32: return launch_epnp_stages<float>(sa, workspace, workspace_bytes, first_round, st);
        case MR_F16: return launch_epnp_stages<__half>(sa, workspace, workspace_bytes, first_round, st);
        case MR_F64: return launch_epnp_stages<double>(sa, workspace, workspace_bytes, first_round, st);
        default: return MR_ERR_UNSUPPORTED;
    }
}

int mr_epnp_ransac_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *ransac_thr, int B, int P,
    float istd_thres, int flags, int max_iters,
    double *init_pose, uint8_t *init_mask, uint8_t *init_valid, float *diag, double *debug_hypotheses,
    void *workspace, size_t workspace_bytes, void *stream) {
    return epnp_ransac_launch(1, &x2d, x2d_strides, &istd, istd_strides, &x3d, x3d_strides, in_dtype, &cam_mats, cam_batch, &ransac_thr, B, P,
                              istd_thres, flags, max_iters, &init_pose, &init_mask, &init_valid, &diag, debug_hypotheses, workspace, workspace_bytes, stream);
}

int mr_epnp_ransac_grouped(
    int ncalls, const void *const *x2d, const int64_t *x2d_strides, const void *const *istd, const int64_t *istd_strides,
    const void *const *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *const *cam_mats, int cam_batch, const float *const *ransac_thr, int B, int P,
    float istd_thres, int flags, int max_iters,
    double *const *init_pose, uint8_t *const *init_mask, uint8_t *const *init_valid, float *const *diag,
    void *workspace, size_t workspace_bytes, void *stream) {
    return epnp_ransac_launch(ncalls, x2d, x2d_strides, istd, istd_strides, x3d, x3d_strides, in_dtype, cam_mats, cam_batch, ransac_thr, B, P,
                              istd_thres, flags, max_iters, init_pose, init_mask, init_valid, diag, nullptr, workspace, workspace_bytes, stream);
}

size_t mr_epnp_workspace_bytes(int B, int P) {
    if (B <= 0 || P < 4) return 0;
    return epnp_work_bytes(B, P, nullptr, nullptr);
}

int mr_cov_symeig_rule(uint8_t *valid, float *cov, int B, float *eig_min_max, void *stream) {
    if (B < 0) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if (!valid || !cov) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(cov_symeig_rule_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, valid, cov, B, eig_min_max);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

int mr_pnp6_refine_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    const uint8_t *inlier_mask, const float *pose4, const uint8_t *valid4, int B, int P, float z_min, int flags,
    uint8_t *valid, float *pose6, float *cov6, float *diag, void *stream) {
    if (B < 0 || P < 4 || P > 64 * kMaxChunks) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if (!x2d || !istd || !x3d || !x2d_strides || !istd_strides || !x3d_strides || !cam_mats || !u_range || !v_range || !inlier_mask || !pose4 ||
        !valid4 || !valid || !pose6 || !cov6)
        return MR_ERR_BAD_ARGUMENT;
    if ((cam_batch != 1 && cam_batch != B) || (range_batch != 1 && range_batch != B)) return MR_ERR_BAD_ARGUMENT;
    Pnp6Args a;
    memset(&a, 0, sizeof a);
    a.x2d = x2d; a.istd = istd; a.x3d = x3d;
    for (int i = 0; i < 3; ++i) { a.s2[i] = x2d_strides[i]; a.sw[i] = istd_strides[i]; a.s3[i] = x3d_strides[i]; }
    a.K = cam_mats; a.K_stride = (cam_batch == 1) ? 0 : 9; a.K_f64 = 0;
    a.ur = u_range; a.vr = v_range; a.r_stride = (range_batch == 1) ? 0 : 2; a.r_f64 = 0;
    a.mask = inlier_mask; a.pose4 = pose4; a.valid4 = valid4;
    a.B = B; a.P = P; a.z_min = (double)z_min;
    { const int mi = (flags & MR_LM_MAXIT_MASK) >> MR_LM_MAXIT_SHIFT; a.lm_max_iter = mi ? mi : 50; }
    a.valid = valid; a.pose6 = pose6; a.cov6 = cov6; a.diag = diag;
    switch (in_dtype) {
        case MR_F32: return launch_pnp6<float>(a, (hipStream_t)stream);
        case MR_F16: return launch_pnp6<__half>(a, (hipStream_t)stream);
        case MR_F64: return launch_pnp6<double>(a, (hipStream_t)stream);
        default: return MR_ERR_UNSUPPORTED;
    }
}

int mr_pnp_exact_hessian_batched(
    const void *x2d, const int64_t *x2d_strides, const void *istd, const int64_t *istd_strides,
    const void *x3d, const int64_t *x3d_strides, int in_dtype,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    const float *pose, const uint8_t *inlier_mask, int B, int P, float z_min,
    uint8_t *valid, float *hess, float *cov, void *stream) {
    if (B < 0 || P < 1) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if (!x2d || !istd || !x3d || !x2d_strides || !istd_strides || !x3d_strides || !cam_mats || !u_range || !v_range || !pose || !valid || !cov)
        return MR_ERR_BAD_ARGUMENT;
    if ((cam_batch != 1 && cam_batch != B) || (range_batch != 1 && range_batch != B)) return MR_ERR_BAD_ARGUMENT;
    HessArgs a;
    memset(&a, 0, sizeof a);
    a.x2d = x2d; a.istd = istd; a.x3d = x3d;
    for (int i = 0; i < 3; ++i) { a.s2[i] = x2d_strides[i]; a.sw[i] = istd_strides[i]; a.s3[i] = x3d_strides[i]; }
    a.K = cam_mats; a.K_stride = (cam_batch == 1) ? 0 : 9;
    a.ur = u_range; a.vr = v_range; a.r_stride = (range_batch == 1) ? 0 : 2;
    a.pose = pose; a.mask = inlier_mask; a.B = B; a.P = P; a.z_min = (double)z_min;
    a.valid = valid; a.hess = hess; a.cov = cov;
    hipStream_t st = (hipStream_t)stream;
    switch (in_dtype) {
        case MR_F32: hipLaunchKernelGGL((exact_hessian_kernel<float>), dim3(B), dim3(256), 0, st, a); break;
        case MR_F16: hipLaunchKernelGGL((exact_hessian_kernel<__half>), dim3(B), dim3(256), 0, st, a); break;
        case MR_F64: hipLaunchKernelGGL((exact_hessian_kernel<double>), dim3(B), dim3(256), 0, st, a); break;
        default: return MR_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

static int fill_decode_args(DecodeArgs &a, const void *all_pred, int pred_dtype, const int64_t *labels, const uint8_t *flip, const float *dim,
                            const float *dim_var, const float *rois, int B, int num_classes, int class_agnostic, int h, int w,
                            const float *dim_means, const float *dim_stds, const float *noc_means, const float *noc_stds,
                            double proj_scaling_denominator, double ref_focal_y, double epistemic_std_gain, float std_scale,
                            float ransac_thres_ratio) {
    if (B < 0 || h < 1 || w < 1 || num_classes < 1) return MR_ERR_BAD_ARGUMENT;
    if (!all_pred || !labels || !flip || !dim || !rois || !dim_means || !dim_stds || !noc_means || !noc_stds) return MR_ERR_BAD_ARGUMENT;
    memset(&a, 0, sizeof a);
    if (pred_dtype != MR_F32 && pred_dtype != MR_F16 && pred_dtype != MR_BF16) return MR_ERR_UNSUPPORTED;
    a.all_pred = all_pred; a.pred_dtype = pred_dtype; a.labels = (const long long *)labels; a.flip = flip; a.dim = dim; a.dim_var = dim_var; a.rois = rois;
    a.B = B; a.C = num_classes; a.agnostic = class_agnostic; a.h = h; a.w = w;
    a.dim_means = dim_means; a.dim_stds = dim_stds;
    a.noc_means = noc_means; a.noc_stds = noc_stds;
    // python-scalar constants of distance_invar_proj_error_coder.py:50-54, rounded the way torch rounds them
    const double e = ref_focal_y * epistemic_std_gain;
    a.k_epi = (float)(e * e);
    a.k_sd2 = (float)(proj_scaling_denominator * proj_scaling_denominator);
    const float sdf = (float)proj_scaling_denominator;
    a.sd_sq = sdf * sdf;
    a.std_scale = std_scale; a.ratio = ransac_thres_ratio; a.has_var = dim_var != nullptr;
    a.w_magic = (w > 1 && (long long)h * w < 65536) ? 0xFFFFFFFFu / (unsigned)w + 1u : 0u;
    return MR_OK;
}

int mr_noc_decode_batched(
    const void *all_pred, int pred_dtype, const int64_t *labels, const uint8_t *flip, const float *dim, const float *dim_var, const float *rois,
    int B, int num_classes, int class_agnostic, int h, int w,
    const float *dim_means, const float *dim_stds, const float *noc_means, const float *noc_stds,
    double proj_scaling_denominator, double ref_focal_y, double epistemic_std_gain, float std_scale, float ransac_thres_ratio,
    float *coords_2d, float *coords_2d_istd, float *coords_3d, float *dims, float *dims_var, float *ransac_thr,
    const float *coord_2d_map, int map_h, int map_w, void *stream) {
    if (B == 0) return MR_OK;
    if (coord_2d_map && (map_h < 1 || map_w < 1)) return MR_ERR_BAD_ARGUMENT;
    DecodeArgs a;
    const int rc = fill_decode_args(a, all_pred, pred_dtype, labels, flip, dim, dim_var, rois, B, num_classes, class_agnostic, h, w, dim_means, dim_stds,
                                    noc_means, noc_stds, proj_scaling_denominator, ref_focal_y, epistemic_std_gain, std_scale, ransac_thres_ratio);
    if (rc != MR_OK) return rc;
    if (!coords_2d || !coords_2d_istd || !coords_3d) return MR_ERR_BAD_ARGUMENT;
    a.c2d = coords_2d; a.istd = coords_2d_istd; a.c3d = coords_3d; a.dims = dims; a.dims_var = dims_var;
    a.thr = (ransac_thres_ratio >= 0.f) ? ransac_thr : nullptr;
    a.map2d = coord_2d_map; a.map_h = map_h; a.map_w = map_w;
    const int hw = h * w;
    const bool x4 = pred_dtype == MR_F32 && !coord_2d_map && (hw % 4 == 0) && a.w_magic != 0u &&      // w_magic: p / w by multiplication (decode_pixel_pair)
                    ((((uintptr_t)all_pred | (uintptr_t)coords_2d | (uintptr_t)coords_2d_istd | (uintptr_t)coords_3d) & 15) == 0);
    if (x4) {
        // 256 threads x one quad measured best (13.1 us per 1024 x 28x28 batch; 128 x 2 quads 14.2, 64 x 4 quads 25.5: the kernel wants threads, not trips);
        // a persistent, three-stage software-pipelined form (loads of the next quad in flight during the arithmetic; bit-identical outputs) is NOT faster:
        // 13.0 - 14.9 us against 12.5 in the same session (tools/ubench/k2_pipelined_experiment.inc, profiles/r04_k2_pipelined_experiment.txt)
#ifdef MR_K2_EXPERIMENT
        {
            static const int lds = getenv("MR_K2_LDS") ? atoi(getenv("MR_K2_LDS")) : 0;       // dynamic LDS nobody uses: caps the workgroups resident per CU
            static const int thr = getenv("MR_K2_THREADS") ? atoi(getenv("MR_K2_THREADS")) : 256;
            if (lds > 48 * 1024) {
                static bool once = false;
                if (!once) { (void)hipFuncSetAttribute((const void *)noc_decode_kernel_x4<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                             (void)hipFuncSetAttribute((const void *)noc_decode_kernel_x4<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                             (void)hipFuncSetAttribute((const void *)noc_decode_kernel_x4<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); once = true; }
            }
            if (thr == 64) hipLaunchKernelGGL((noc_decode_kernel_x4<64, 4>), dim3((unsigned)B), dim3(64), lds, (hipStream_t)stream, a, hw / 4, g_stamps);
            else if (thr == 128) hipLaunchKernelGGL((noc_decode_kernel_x4<128, 2>), dim3((unsigned)B), dim3(128), lds, (hipStream_t)stream, a, hw / 4, g_stamps);
            else hipLaunchKernelGGL((noc_decode_kernel_x4<256, 1>), dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, a, hw / 4, g_stamps);
            HIP_TRY(hipGetLastError());
            return MR_OK;
        }
#else
        hipLaunchKernelGGL((noc_decode_kernel_x4<256, 1>), dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a, hw / 4);
        HIP_TRY(hipGetLastError());
        return MR_OK;
#endif
    }
    const long long blocks = (long long)((hw + 255) / 256) * B;
    if (blocks > 0x7fffffffLL) return MR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(noc_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

int mr_pnp_from_head_batched(
    const void *all_pred, int pred_dtype, const int64_t *labels, const uint8_t *flip, const float *dim, const float *dim_var, const float *rois,
    int B, int num_classes, int class_agnostic, int h, int w,
    const float *dim_means, const float *dim_stds, const float *noc_means, const float *noc_stds,
    double proj_scaling_denominator, double ref_focal_y, double epistemic_std_gain, float std_scale, float ransac_thres_ratio,
    const float *cam_mats, int cam_batch, const float *u_range, const float *v_range, int range_batch,
    float z_min, float istd_thres, int inlier_opt_only, int flags,
    uint8_t *valid, float *pose, float *cov, float *tr_radius, uint8_t *inlier_mask, float *diag,
    float *dims, float *dims_var, const float *coord_2d_map, int map_h, int map_w,
    const float *cov_calib_logscale, float cov_corr_sd, float *cov_calib, void *stream) {
    const int P = h * w;
    if (B < 0 || P < 4 || P > 64 * kMaxChunks) return MR_ERR_BAD_ARGUMENT;
    if (coord_2d_map && (map_h < 1 || map_w < 1)) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if (!cam_mats || !u_range || !v_range || !valid || !pose || !tr_radius || (!cov && !(flags & MR_COV_NONE))) return MR_ERR_BAD_ARGUMENT;
    if ((cam_batch != 1 && cam_batch != B) || (range_batch != 1 && range_batch != B)) return MR_ERR_BAD_ARGUMENT;
    PnpArgs a;
    memset(&a, 0, sizeof a);
    const int rc = fill_decode_args(a.dec, all_pred, pred_dtype, labels, flip, dim, dim_var, rois, B, num_classes, class_agnostic, h, w, dim_means, dim_stds,
                                    noc_means, noc_stds, proj_scaling_denominator, ref_focal_y, epistemic_std_gain, std_scale, ransac_thres_ratio);
    if (rc != MR_OK) return rc;
    a.dec.dims = dims; a.dec.dims_var = dims_var;
    a.dec.map2d = coord_2d_map; a.dec.map_h = map_h; a.dec.map_w = map_w;
    if (cov_calib && (!cov_calib_logscale || (flags & MR_COV_NONE))) return MR_ERR_BAD_ARGUMENT;
    a.calib_logscale = cov_calib_logscale; a.corr_sd = cov_corr_sd; a.cov_calib = cov_calib;
    a.from_head = 1;
    // the tile is built channel-planar, exactly the layout (and hence numpy summation order) the reference's head produces
    a.s2[0] = 2LL * P; a.s2[1] = 1; a.s2[2] = P; a.sw[0] = 2LL * P; a.sw[1] = 1; a.sw[2] = P; a.s3[0] = 3LL * P; a.s3[1] = 1; a.s3[2] = P;
    a.K = cam_mats; a.K_stride = (cam_batch == 1) ? 0 : 9; a.K_f64 = 0;
    a.ur = u_range; a.vr = v_range; a.r_stride = (range_batch == 1) ? 0 : 2; a.r_f64 = 0;
    a.B = B; a.P = P; a.z_min = (double)z_min; a.istd_thres = istd_thres; a.inlier_opt_only = inlier_opt_only; a.flags = flags;
    a.valid = valid; a.pose = pose; a.cov = cov; a.tr = tr_radius; a.mask = inlier_mask; a.diag = diag;
    a.stamps = g_stamps;
    int mm = flags & MR_MEAN_MASK;
    if (mm == MR_MEAN_AUTO) mm = MR_MEAN_PAIRWISE;
    a.mean_mode = mm;
    if (mm == MR_MEAN_PAIRWISE && !(flags & MR_NO_ISTD_MASK)) {
        if (!build_plan(a.plan, P)) return MR_ERR_UNSUPPORTED;
    }
    return launch_wpo<float>(a, widen_for_large_tiles(pick_wpo(B, P, flags), a, flags, MR_F32), (hipStream_t)stream);
}

int mr_roi_align_avg(const float *input, const float *rois, int K, int C, int H, int W, int out_h, int out_w,
                     float spatial_scale, int sampling_ratio, int aligned, float *output, void *stream) {
    if (K < 0 || C < 1 || H < 1 || W < 1 || out_h < 1 || out_w < 1) return MR_ERR_BAD_ARGUMENT;
    if (K == 0) return MR_OK;
    if (!input || !rois || !output) return MR_ERR_BAD_ARGUMENT;
    const long long n = (long long)K * C * out_h * out_w, blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return MR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(roi_align_avg_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, input, rois, K, C, H, W, out_h, out_w,
                       spatial_scale, sampling_ratio, aligned, output);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

int mr_nms_bev_batched(const float *boxes_xyxyr, const float *scores, const int32_t *offsets, int groups, int max_group,
                       float thr, int64_t *keep, int32_t *num_keep, void *stream) {
    if (groups < 0 || max_group < 0) return MR_ERR_BAD_ARGUMENT;
    if (groups == 0) return MR_OK;
    if (!offsets || !keep || !num_keep || (max_group > 0 && (!boxes_xyxyr || !scores))) return MR_ERR_BAD_ARGUMENT;
    if (max_group > kNmsMax) return MR_ERR_UNSUPPORTED;
    int np2 = 1; while (np2 < max_group) np2 <<= 1;
    const size_t lds = (size_t)np2 * 8 + (size_t)max_group * sizeof(NmsBox) + (size_t)max_group * ((max_group + 31) / 32) * 4 + 16;
    hipLaunchKernelGGL(nms_bev_kernel, dim3(groups), dim3(256), lds, (hipStream_t)stream, boxes_xyxyr, scores, (const int *)offsets, thr,
                       (long long *)keep, (int *)num_keep);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

// ---- N2: KITTI evaluator (eval.py / rotate_iou.py of core/evaluation/kitti_utils)
int mr_kitti_overlaps(int metric, int arith32, int out32, int n_img, const int64_t *dt_off, const int64_t *gt_off, const int64_t *ov_off,
                      int64_t total_pairs, const double *dt_box, const double *gt_box, double *overlaps, void *stream) {
    if (metric < 0 || metric > 2 || n_img < 0 || total_pairs < 0) return MR_ERR_BAD_ARGUMENT;
    if (n_img == 0 || total_pairs == 0) return MR_OK;
    if (!dt_off || !gt_off || !ov_off || !dt_box || !gt_box || !overlaps) return MR_ERR_BAD_ARGUMENT;
    const long long blocks = (total_pairs + 255) / 256;
    if (blocks > 0x7fffffffLL) return MR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kitti_overlap_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, metric, arith32, out32, n_img,
                       (const long long *)dt_off, (const long long *)gt_off, (const long long *)ov_off, dt_box, gt_box, overlaps);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

int64_t mr_kitti_match_workspace_bytes(int n_img, int n_combo) {
    if (n_img < 0 || n_combo < 0) return 0;
    return (int64_t)n_combo * kEvalSamples * (int64_t)n_img * (3 * sizeof(int) + sizeof(double)) + 64;
}

int mr_kitti_match(int second_pass, int metric, int compute_aos, int alpha32, int dtdata32, int n_img, int max_det,
                   const int64_t *dt_off, const int64_t *gt_off, const int64_t *ov_off, const int64_t *dc_off,
                   int64_t total_dt, int64_t total_gt,
                   const double *overlaps, const double *dt_box, const double *dt_alpha, const double *gt_alpha, const double *dc_box,
                   const int8_t *ign_gt, const int8_t *ign_dt, int n_combo, const int32_t *combo_cd, const double *combo_min_overlap,
                   const double *thresholds, const int32_t *n_thr, double *match_score, double *pr,
                   void *workspace, int64_t workspace_bytes, void *stream) {
    if (n_img < 0 || n_combo < 0 || metric < 0 || metric > 2) return MR_ERR_BAD_ARGUMENT;
    if (n_img == 0 || n_combo == 0) return MR_OK;
    if (max_det > kEvalMaxDet) return MR_ERR_UNSUPPORTED;
    if (!dt_off || !gt_off || !ov_off || !dc_off || !ign_gt || !ign_dt || !combo_cd || !combo_min_overlap) return MR_ERR_BAD_ARGUMENT;
    MatchArgs a;
    a.second_pass = second_pass; a.metric = metric; a.compute_aos = compute_aos; a.alpha32 = alpha32; a.dtdata32 = dtdata32;
    a.n_img = n_img; a.n_combo = n_combo; a.total_gt = total_gt; a.total_dt = total_dt;
    a.dt_off = (const long long *)dt_off; a.gt_off = (const long long *)gt_off; a.ov_off = (const long long *)ov_off; a.dc_off = (const long long *)dc_off;
    a.ov = overlaps; a.dt_box = dt_box; a.dt_alpha = dt_alpha; a.gt_alpha = gt_alpha; a.dc_box = dc_box;
    a.ign_gt = (const signed char *)ign_gt; a.ign_dt = (const signed char *)ign_dt;
    a.combo_cd = (const int *)combo_cd; a.combo_min_overlap = combo_min_overlap;
    a.thresholds = thresholds; a.n_thr = (const int *)n_thr; a.match_score = match_score;
    a.st_tp = a.st_fp = a.st_fn = nullptr; a.st_sim = nullptr;
    long long threads = (long long)n_combo * n_img;
    if (second_pass) {
        if (!thresholds || !n_thr || !pr || !workspace || workspace_bytes < mr_kitti_match_workspace_bytes(n_img, n_combo)) return MR_ERR_BAD_ARGUMENT;
        const long long cells = (long long)n_combo * kEvalSamples * n_img;
        a.st_sim = (double *)workspace;                       // doubles first (alignment), then the three int planes
        a.st_tp = (int *)(a.st_sim + cells); a.st_fp = a.st_tp + cells; a.st_fn = a.st_fp + cells;
        threads *= kEvalSamples;
    } else if (!match_score) return MR_ERR_BAD_ARGUMENT;
    const long long blocks = (threads + 127) / 128;
    if (blocks > 0x7fffffffLL) return MR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kitti_match_kernel, dim3((unsigned)blocks), dim3(128), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    if (second_pass) {
        hipLaunchKernelGGL(kitti_reduce_kernel, dim3((n_combo * kEvalSamples + 63) / 64), dim3(64), 0, (hipStream_t)stream, n_img, n_combo,
                           (const int *)n_thr, a.st_tp, a.st_fp, a.st_fn, a.st_sim, pr);
        HIP_TRY(hipGetLastError());
    }
    return MR_OK;
}

// Batched form of the two 7-parameter solvers (device fp64 buffers, one workgroup per object; pnp_noc_kernel.inc)
int mr_pnp_noc_batched(int full_cov, const double *pts2d, const double *pts3d, const double *wgt2d, const double *logdim, const double *logdim_wgt,
                       const double *K, int K_batch, const double *init_dimpose, const double *clips, int clips_batch, double delta, int B, int pn,
                       double *result_dimpose, int32_t *result_val, double *diag, void *stream) {
    if (B < 0 || pn < 0 || (K_batch != 1 && K_batch != B) || (clips_batch != 1 && clips_batch != B)) return MR_ERR_BAD_ARGUMENT;
    if (B == 0) return MR_OK;
    if ((pn > 0 && (!pts2d || !pts3d || !wgt2d)) || !logdim || !logdim_wgt || !K || !init_dimpose || !clips || !result_dimpose || !result_val) return MR_ERR_BAD_ARGUMENT;
    NocArgs a;
    memset(&a, 0, sizeof a);
    a.pts2d = pts2d; a.pts3d = pts3d; a.wgt2d = wgt2d; a.logdim = logdim; a.logdim_wgt = logdim_wgt; a.K = K; a.init = init_dimpose; a.clips = clips;
    a.K_batch = K_batch; a.clips_batch = clips_batch; a.delta = delta; a.pn = pn; a.full_cov = full_cov ? 1 : 0; a.B = B;
    a.out_dimpose = result_dimpose; a.out_val = (int *)result_val; a.out_diag = diag;
    hipLaunchKernelGGL(pnp_noc_kernel, dim3(B), dim3(256), kNocLds, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

// ---- host-buffer entry points of the reference's C ABI (ext.h).  Per device: one private non-blocking stream, one pinned
// host staging buffer and one device buffer, grown on demand and kept; a call is one async H2D copy, the kernel and one async
// D2H copy on that stream followed by a single hipStreamSynchronize (no default-stream launch, no pageable copies, no
// allocation in the steady state).  Calls on the same device serialise on the stage's mutex (the reference invokes these
// serially, pnp_uncert_cpu.py:180-191); calls on different devices run concurrently.
struct HostStage {
    std::mutex mu;
    hipStream_t st = nullptr;
    void *dbuf = nullptr, *hbuf = nullptr;
    size_t cap = 0;
};
static HostStage g_stage[kMaxDevices];

// returns the locked stage of the current device with room for `bytes` in both buffers, or nullptr (lock not held)
static HostStage *stage_acquire(size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { g_last_hip_error = (int)hipGetLastError(); return nullptr; }
    HostStage *s = &g_stage[dev];
    s->mu.lock();
    bool ok = true;
    if (!s->st) ok = hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) == hipSuccess;
    if (ok && bytes > s->cap) {
        const size_t want = bytes < 4096 ? 4096 : bytes + bytes / 2;
        if (s->dbuf) (void)hipFree(s->dbuf);
        if (s->hbuf) (void)hipHostFree(s->hbuf);
        s->dbuf = s->hbuf = nullptr; s->cap = 0;
        ok = hipMalloc(&s->dbuf, want) == hipSuccess && hipHostMalloc(&s->hbuf, want, hipHostMallocDefault) == hipSuccess;
        if (ok) s->cap = want;
    }
    if (!ok) { g_last_hip_error = (int)hipGetLastError(); s->mu.unlock(); return nullptr; }
    return s;
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
    // staging, in doubles: [pts2d 2P | pts3d 3P | wgt 2P | K 9 | ur 2 | vr 2 | init 4 || pose64 4 | cov64 16 | tr64 1 | valid (u8, 8 bytes)]
    //                      + [pose 4 | cov 16 | tr 1] floats (written by the kernel, not read back)
    const size_t nin = (size_t)7 * P + 9 + 2 + 2 + 4, nout = 4 + 16 + 1 + 1;
    const size_t bytes = (nin + nout) * sizeof(double) + 24 * sizeof(float);
    HostStage *sg = stage_acquire(bytes);
    if (!sg) return;
    std::lock_guard<std::mutex> lk(sg->mu, std::adopt_lock);
    double *h = (double *)sg->hbuf, *d = (double *)sg->dbuf;
    double *h2 = h, *h3 = h2 + 2 * P, *hw = h3 + 3 * P, *hK = hw + 2 * P, *hur = hK + 9, *hvr = hur + 2, *hin = hvr + 2;
    memcpy(h2, pts2d, sizeof(double) * 2 * pn); memcpy(h3, pts3d, sizeof(double) * 3 * pn); memcpy(hw, wgt2d, sizeof(double) * 2 * pn);
    for (int p = pn; p < P; ++p) { h2[2 * p] = h2[2 * p + 1] = 0.0; h3[3 * p] = h3[3 * p + 1] = 0.0; h3[3 * p + 2] = 1.0; hw[2 * p] = hw[2 * p + 1] = 0.0; }   // padded points carry zero weight
    memcpy(hK, K, sizeof(double) * 9);
    hur[0] = clips[1]; hur[1] = clips[2]; hvr[0] = clips[3]; hvr[1] = clips[4];
    memcpy(hin, init_pose, sizeof(double) * 4);
    if (hipMemcpyAsync(d, h, nin * sizeof(double), hipMemcpyHostToDevice, sg->st) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); return; }
    PnpArgs a;
    memset(&a, 0, sizeof a);
    a.x2d = d; a.x3d = d + 2 * P; a.istd = d + 5 * P;
    a.s2[0] = 0; a.s2[1] = 2; a.s2[2] = 1; a.sw[0] = 0; a.sw[1] = 2; a.sw[2] = 1; a.s3[0] = 0; a.s3[1] = 3; a.s3[2] = 1;
    a.K = d + 7 * P; a.K_stride = 0; a.K_f64 = 1;
    a.ur = d + 7 * P + 9; a.vr = d + 7 * P + 11; a.r_stride = 0; a.r_f64 = 1;
    a.init_pose = d + 7 * P + 13;
    double *dout = d + nin;
    a.pose64 = dout; a.cov64 = dout + 4; a.tr64 = dout + 20; a.valid = (uint8_t *)(dout + 21);
    float *df = (float *)(dout + nout);
    a.pose = df; a.cov = df + 4; a.tr = df + 20;
    a.B = 1; a.P = P; a.z_min = clips[0]; a.istd_thres = 0.f; a.inlier_opt_only = 0;
    a.flags = MR_NO_ISTD_MASK | (result_cov ? MR_COV_CERES : MR_COV_NONE);
    a.mean_mode = MR_MEAN_SEQUENTIAL;
    int wpo = 1; while (wpo < 8 && P >= 64 * wpo * 2) wpo *= 2;
    if (launch_wpo<double>(a, wpo, sg->st) != MR_OK) return;
    double *ho = h + nin;
    if (hipMemcpyAsync(ho, dout, nout * sizeof(double), hipMemcpyDeviceToHost, sg->st) != hipSuccess ||
        hipStreamSynchronize(sg->st) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); return; }
    const uint8_t ok = *(const uint8_t *)(ho + 21);
    memcpy(result_pose, ho, 4 * sizeof(double));
    *result_tr = ho[20];
    *result_val = ok ? 1 : 0;
    if (ok && result_cov) memcpy(result_cov, ho + 4, 16 * sizeof(double));
}

// The 7-parameter entry points of the reference's C ABI (ext.h:15-43).  Host fp64 buffers; one object; blocking.
static void noc_host(int full_cov, double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt, double *K,
                     double *init_dimpose, int *result_val, double *result_dimpose, int pn, double *clips, double delta) {
    *result_val = 0;
    memcpy(result_dimpose, init_dimpose, 7 * sizeof(double));            // pnp_uncert_cpu.cpp:309,351
    if (pn < 0) return;
    const int ws = full_cov ? 3 : 2;
    // staging, in doubles: [pts2d 2n | pts3d 3n | wgt ws*n | logdim 3 | logdim_wgt 3 | K 9 | init 7 | clips 5 || out 7 | val (int, 8 bytes)]
    const size_t n = (size_t)pn;
    const size_t nin = (2 + 3 + ws) * n + 3 + 3 + 9 + 7 + 5, nout = 7 + 1;
    HostStage *sg = stage_acquire((nin + nout) * sizeof(double));
    if (!sg) return;
    std::lock_guard<std::mutex> lk(sg->mu, std::adopt_lock);
    double *h = (double *)sg->hbuf, *d = (double *)sg->dbuf;
    double *q = h;
    memcpy(q, pts2d, sizeof(double) * 2 * n); q += 2 * n;
    memcpy(q, pts3d, sizeof(double) * 3 * n); q += 3 * n;
    memcpy(q, wgt2d, sizeof(double) * ws * n); q += ws * n;
    memcpy(q, logdim, sizeof(double) * 3); q += 3;
    memcpy(q, logdim_wgt, sizeof(double) * 3); q += 3;
    memcpy(q, K, sizeof(double) * 9); q += 9;
    memcpy(q, init_dimpose, sizeof(double) * 7); q += 7;
    memcpy(q, clips, sizeof(double) * 5);
    if (hipMemcpyAsync(d, h, nin * sizeof(double), hipMemcpyHostToDevice, sg->st) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); return; }
    NocArgs a;
    memset(&a, 0, sizeof a);
    a.pts2d = d; a.pts3d = d + 2 * n; a.wgt2d = d + 5 * n; a.logdim = d + (5 + ws) * n; a.logdim_wgt = a.logdim + 3; a.K = a.logdim + 6;
    a.init = a.logdim + 15; a.clips = a.logdim + 22; a.out_dimpose = d + nin; a.out_val = (int *)(d + nin + 7); a.out_diag = nullptr;
    a.K_batch = 1; a.clips_batch = 1; a.B = 1;
    a.delta = delta; a.pn = pn; a.full_cov = full_cov;
    hipLaunchKernelGGL(pnp_noc_kernel, dim3(1), dim3(256), kNocLds, sg->st, a);
    if (hipGetLastError() != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); return; }
    double *ho = h + nin;
    if (hipMemcpyAsync(ho, d + nin, nout * sizeof(double), hipMemcpyDeviceToHost, sg->st) != hipSuccess ||
        hipStreamSynchronize(sg->st) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); return; }
    memcpy(result_dimpose, ho, 7 * sizeof(double));
    *result_val = *(const int *)(ho + 7);
}

void pnp_noc_uncert(double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt, double *K,
                    double *init_dimpose, int *result_val, double *result_dimpose, int pn, double *clips, double delta) {
    noc_host(0, pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, init_dimpose, result_val, result_dimpose, pn, clips, delta);
}

void pnp_noc_cov_uncert(double *pts2d, double *pts3d, double *wgt2d, double *logdim, double *logdim_wgt, double *K,
                        double *init_dimpose, int *result_val, double *result_dimpose, int pn, double *clips, double delta) {
    noc_host(1, pts2d, pts3d, wgt2d, logdim, logdim_wgt, K, init_dimpose, result_val, result_dimpose, pn, clips, delta);
}

}  // extern "C"

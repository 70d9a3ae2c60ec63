// Microbenchmark: latency of the operations on the LM leader's critical path on gfx950, for a wave that is alone on its SIMD
// (1024 workgroups x 64 threads = one wave per SIMD): cycles per operation in a dependent chain vs with 2 / 4 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double dpp_shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rdlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, int iters) {
    __shared__ double sh[512];
    double a0 = threadIdx.x * 1e-3 + 1.5, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float f0 = (float)a0;
    const double m = 1.0000001, c = 1e-9;
    sh[threadIdx.x] = (double)((threadIdx.x * 17 + 5) & 63);
    sh[threadIdx.x + 64] = a0;
    __syncthreads();
    int idx = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { a0 = fma(a0, m, c); }
            if (MODE == 1) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); }
            if (MODE == 2) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); }
            if (MODE == 3) { a0 = __builtin_amdgcn_rcp(a0) + 1.0; }                       // rcp + add
            if (MODE == 4) { a0 = __builtin_amdgcn_rsq(a0) + 1.0; }
            if (MODE == 5) { a0 = sqrt(a0) + 1.0; }                                       // IEEE sqrt
            if (MODE == 6) { a0 = 1.0 / a0 + 1.0; }                                       // IEEE div
            if (MODE == 7) { a0 = dpp_shr1(a0) + c; }                                     // DPP pair + add
            if (MODE == 8) { a0 = rdlane(a0, 3) * a1 + c; }                               // readlane pair -> SGPR operand
            if (MODE == 9) { idx = (int)sh[idx]; }                                        // LDS pointer chase: ds_read_b64 + cvt
            if (MODE == 10) { f0 = fmaf(f0, 1.0000001f, 1e-9f); }
            if (MODE == 11) { a0 = fma(a0, m, c); a0 = (double)(float)a0; }               // fma + cvt pair
            if (MODE == 12) { sh[64 + threadIdx.x] = a0; __syncthreads(); a0 = sh[64 + ((threadIdx.x + 64) & (blockDim.x - 1))] + c; }  // LDS write, barrier, read
            if (MODE == 13) { a0 = fmax(a0 * m, c); }                                     // mul + max
            if (MODE == 14) { a0 = (a0 < a1) ? a0 + c : a0 - c; }                         // compare + 2 ops + select
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + idx;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int threads, double ops) {
    const int blocks = 1024;
    double *o; unsigned long long *c; hipMalloc(&o, sizeof(double) * blocks * threads); hipMalloc(&c, 8 * blocks);
    const int iters = 512;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, o, c, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), c, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v; s /= blocks;
    printf("%-44s threads=%3d : %7.1f cycles per step (%.1f per op)\n", name, threads, s / iters / 8, s / iters / 8 / ops);
    hipFree(o); hipFree(c);
}
int main() {
    run<0>("fma_f64 dependent", 64, 1);
    run<1>("fma_f64 2 chains", 64, 2);
    run<2>("fma_f64 4 chains", 64, 4);
    run<10>("fma_f32 dependent", 64, 1);
    run<3>("v_rcp_f64 + add (dependent)", 64, 2);
    run<4>("v_rsq_f64 + add (dependent)", 64, 2);
    run<5>("IEEE sqrt f64 + add (dependent)", 64, 1);
    run<6>("IEEE div f64 + add (dependent)", 64, 1);
    run<7>("DPP row_shr:1 (2 movs) + add", 64, 1);
    run<8>("readlane x2 -> fma with SGPR operand", 64, 1);
    run<9>("LDS pointer chase (ds_read_b64 + cvt)", 64, 1);
    run<11>("fma + cvt f64->f32->f64", 64, 3);
    run<13>("mul + max", 64, 2);
    run<14>("cmp + add/sub + select", 64, 1);
    run<12>("LDS write + barrier + read, 1 wave", 64, 1);
    run<12>("LDS write + barrier + read, 4 waves", 256, 1);
    return 0;
}

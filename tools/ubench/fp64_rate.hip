// Microbenchmark: cycles per wave-instruction for fp64 / fp32 VALU ops on gfx950 (one wave per SIMD and 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, int iters) {
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
    const double m = 1.0000001, c = 1e-9; const float mf = 1.0000001f, cf = 1e-9f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c); }
        if (MODE == 1) { a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m; }
        if (MODE == 2) { a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c; }
        if (MODE == 3) { f0 = fmaf(f0, mf, cf); f1 = fmaf(f1, mf, cf); f2 = fmaf(f2, mf, cf); f3 = fmaf(f3, mf, cf); f4 = fmaf(f4, mf, cf); f5 = fmaf(f5, mf, cf); f6 = fmaf(f6, mf, cf); f7 = fmaf(f7, mf, cf); }
        if (MODE == 4) { a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); }   // dependent chain
        if (MODE == 5) { a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3); a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7); }
        if (MODE == 6) { a0 = (double)(float)a0; a1 = (double)(float)a1; a2 = (double)(float)a2; a3 = (double)(float)a3; a4 = (double)(float)a4; a5 = (double)(float)a5; a6 = (double)(float)a6; a7 = (double)(float)a7; }  // cvt pair
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int blocks, int threads, int per_iter) {
    double *o; unsigned long long *c; hipMalloc(&o, sizeof(double) * blocks * threads); hipMalloc(&c, 8 * blocks);
    const int iters = 4096;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, o, c, iters); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, o, c, iters); hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), c, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v; s /= blocks;
    printf("%-28s blocks=%5d threads=%4d : %.2f cycles per wave-instruction (per wave)\n", name, blocks, threads, s / iters / per_iter);
    hipFree(o); hipFree(c);
}
int main() {
    for (int cfg = 0; cfg < 3; ++cfg) {
        int blocks = cfg == 0 ? 256 : 1024, threads = cfg == 2 ? 128 : 64;   // 1 wave/CU ; 1 wave/SIMD ; 2 waves/SIMD
        printf("--- %s\n", cfg == 0 ? "1 wave per CU" : cfg == 1 ? "1 wave per SIMD" : "2 waves per SIMD (cycles are per wave)");
        run<0>("v_fma_f64 x8 independent", blocks, threads, 8);
        run<1>("v_mul_f64 x8 independent", blocks, threads, 8);
        run<2>("v_add_f64 x8 independent", blocks, threads, 8);
        run<3>("v_fma_f32 x8 independent", blocks, threads, 8);
        run<4>("v_fma_f64 dependent chain", blocks, threads, 8);
        run<5>("v_rcp_f64 x8 independent", blocks, threads, 8);
        run<6>("cvt f64->f32->f64 x8", blocks, threads, 16);
    }
    return 0;
}

// Development aid: how many QL sweeps / rotations do the EPnP eigen-problems take per matrix (distribution over 30 x 1024 rank-10 Gram
// matrices)?  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/eig12_stats.hip -o tools/ubench/eig12_stats
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#define MR_EXACT _Pragma("clang fp contract(off)")
#define EQ_STATS 1
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
#include "epnp_eig_lanes_round3.inc"
int main(int argc, char **argv) {
    const int nprob = 30 * 1024, rank = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<double> h((size_t)nprob * 144);
    srand(1);
    for (int p = 0; p < nprob; ++p) {
        double M[24][12];
        for (int i = 0; i < rank; ++i) for (double &x : M[i]) x = (rand() / (double)RAND_MAX - 0.5) * 100.0;
        for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) { double s = 0; for (int i = 0; i < rank; ++i) s += M[i][a] * M[i][b]; h[(size_t)p * 144 + 12 * a + b] = s; }
    }
    double *dm, *de, *dw;
    hipMalloc(&dm, h.size() * 8); hipMalloc(&de, (size_t)nprob * 48 * 8); hipMalloc(&dw, (size_t)nprob * 4 * 8);
    hipMemcpy(dm, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> w((size_t)nprob * 4);
    auto clocks = [&](const char *name, int np) {
        hipMemcpy(w.data(), dw, w.size() * 8, hipMemcpyDeviceToHost);
        double cs = 0, ts = 0, cmax = 0, tmax = 0;
        for (int p = 0; p < np; ++p) { cs += w[p * 4 + 2]; ts += w[p * 4 + 3]; cmax = std::max(cmax, w[p * 4 + 2]); tmax = std::max(tmax, w[p * 4 + 3]); }
        printf("%s: per matrix-group mean %.0f shader clocks = %.1f us (mean clock %.0f MHz); slowest %.0f clocks = %.1f us\n", name, cs / np, ts / np * 0.01, cs / ts * 100.0, cmax, tmax * 0.01);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((eig12_lanes_kernel<4, 1>), dim3(16), dim3(64), 0, 0, dm, de, dw, 16); hipDeviceSynchronize(); if (rep) clocks("<4,1> x 16 waves", 16);
        hipLaunchKernelGGL((eig12_lanes_kernel<4, 1>), dim3(256), dim3(64), 0, 0, dm, de, dw, 256); hipDeviceSynchronize(); if (rep) clocks("<4,1> x 256 waves", 256);
        hipLaunchKernelGGL((eig12_lanes_kernel<4, 1>), dim3(1024), dim3(64), 0, 0, dm, de, dw, 1024); hipDeviceSynchronize(); if (rep) clocks("<4,1> x 1024 waves", 1024);
        hipLaunchKernelGGL((eig12_lanes_kernel<4, 15>), dim3(1), dim3(64), 0, 0, dm, de, dw, 15); hipDeviceSynchronize(); if (rep) clocks("<4,15> x 1 wave", 15);
        hipLaunchKernelGGL((eig12_lanes_kernel<4, 15>), dim3(69), dim3(64), 0, 0, dm, de, dw, 1024); hipDeviceSynchronize(); if (rep) clocks("<4,15> x 69 waves", 1024);
        hipLaunchKernelGGL((eig12_lanes_kernel<4, 15>), dim3(1024), dim3(64), 0, 0, dm, de, dw, 15360); hipDeviceSynchronize(); if (rep) clocks("<4,15> x 1024 waves", 15360);
    }
    hipLaunchKernelGGL((eig12_lanes_kernel<4, 15>), dim3((nprob + 14) / 15), dim3(64), 0, 0, dm, de, dw, nprob);
    hipMemcpy(w.data(), dw, w.size() * 8, hipMemcpyDeviceToHost);
    std::vector<int> sw(nprob), ro(nprob);
    for (int p = 0; p < nprob; ++p) { sw[p] = (int)w[p * 4]; ro[p] = (int)w[p * 4 + 1]; }
    std::vector<int> s2 = sw, r2 = ro; std::sort(s2.begin(), s2.end()); std::sort(r2.begin(), r2.end());
    auto pc = [&](std::vector<int> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
    printf("rank %d: sweeps p50 %d p90 %d p99 %d p99.9 %d max %d | rotations p50 %d p90 %d p99 %d p99.9 %d max %d\n", rank, pc(s2, .5), pc(s2, .9), pc(s2, .99), pc(s2, .999), s2.back(),
           pc(r2, .5), pc(r2, .9), pc(r2, .99), pc(r2, .999), r2.back());
    int worst = (int)(std::max_element(ro.begin(), ro.end()) - ro.begin());
    printf("worst matrix %d: sweeps %d rotations %d\n", worst, sw[worst], ro[worst]);
    return 0;
}

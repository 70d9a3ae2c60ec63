// Microbenchmark (development aid): the library's 12x12 eigen-solver (monorun_amd/csrc/epnp_eig_lanes.inc) alone, on the
// 30 x 1024 problems of one EPnP / RANSAC launch, with two and four lanes per matrix.  Prints the launch times and the null-vector
// residual of a few problems; the two mappings must agree bit for bit (same arithmetic specification).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/eig12_lanes.hip -o tools/ubench/eig12_lanes && tools/ubench/eig12_lanes [nprob]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#define MR_EXACT _Pragma("clang fp contract(off)")
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
#include "epnp_eig_lanes_round3.inc"

template <int LPM, int GPW>
static void run(const char *name, const double *dm, double *de, int nprob, std::vector<double> &ev) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((eig12_lanes_kernel<LPM, GPW>), dim3((nprob + GPW - 1) / GPW), dim3(64), 0, 0, dm, de, (double *)nullptr, nprob);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%s: %d problems in %.1f us\n", name, nprob, ms * 1e3);
    }
    hipMemcpy(ev.data(), de, ev.size() * 8, hipMemcpyDeviceToHost);
}

int main(int argc, char **argv) {
    const int nprob = argc > 1 ? atoi(argv[1]) : 30 * 1024;
    std::vector<double> h((size_t)nprob * 144);
    srand(1);
    for (int p = 0; p < nprob; ++p) {           // rank-10 Gram matrices like the 5-point M^T M
        double M[10][12];
        for (auto &r : M) for (double &x : r) x = (rand() / (double)RAND_MAX - 0.5) * 100.0;
        for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) { double s = 0; for (int i = 0; i < 10; ++i) s += M[i][a] * M[i][b]; h[(size_t)p * 144 + 12 * a + b] = s; }
    }
    double *dm, *de;
    hipMalloc(&dm, h.size() * 8); hipMalloc(&de, (size_t)nprob * 48 * 8);
    hipMemcpy(dm, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> ev4((size_t)nprob * 48), ev2(ev4.size()), ev1(ev4.size());
    run<4, 15>("4 lanes per matrix, 15 per wave", dm, de, nprob, ev4);
    run<2, 30>("2 lanes per matrix, 30 per wave", dm, de, nprob, ev2);
    run<2, 20>("2 lanes per matrix, 20 per wave", dm, de, nprob, ev1);
    run<2, 15>("2 lanes per matrix, 15 per wave", dm, de, nprob, ev1);
    run<2, 10>("2 lanes per matrix, 10 per wave", dm, de, nprob, ev1);
    run<4, 8>("4 lanes per matrix, 8 per wave", dm, de, nprob, ev1);
    run<4, 4>("4 lanes per matrix, 4 per wave", dm, de, nprob, ev1);
    run<4, 2>("4 lanes per matrix, 2 per wave", dm, de, nprob, ev1);
    run<4, 1>("4 lanes per matrix, 1 per wave", dm, de, nprob, ev1);
    printf("2-lane and 4-lane results %s\n", memcmp(ev4.data(), ev2.data(), ev4.size() * 8) ? "DIFFER" : "identical (bitwise)");
    double worst = 0;                           // residual of the smallest eigenvector of a few problems: |A v| / |A|
    for (int p = 0; p < 64 && p < nprob; ++p) {
        double r = 0, nrm = 0;
        for (int a = 0; a < 12; ++a) { double s = 0; for (int b = 0; b < 12; ++b) { s += h[(size_t)p * 144 + 12 * a + b] * ev2[(size_t)p * 48 + b]; nrm = fmax(nrm, fabs(h[(size_t)p * 144 + 12 * a + b])); } r = fmax(r, fabs(s)); }
        worst = fmax(worst, r / nrm);
    }
    printf("null-vector residual |A v| / |A| (64 problems): %.2e\n", worst);
    return 0;
}

// Microbenchmark (development aid): the library's 12x12 solver since round 4 (monorun_amd/csrc/epnp_eig_low4.inc: the four smallest
// eigenvectors by tridiagonalisation + bisection + inverse iteration, one quad per matrix) alone, on rank-10 Gram matrices like the
// five-point M^T M.  Prints launch times for several matrices-per-wave settings and writes inputs + results to a file so that the
// CPU restatement (oracle.eig12_low4) can be compared bit for bit (tests/sweeps/check_eig12_low4.py).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/eig12_low4.hip -o tools/ubench/eig12_low4 && tools/ubench/eig12_low4 [nprob] [dump-file]
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
#include "../../monorun_amd/csrc/epnp_eig_low4.inc"

template <int QPW>
__global__ void __launch_bounds__(64, 1) eig_kernel(const double *__restrict__ mats, double *__restrict__ ev, double *__restrict__ w, int nprob) {
    const int grp = (int)threadIdx.x >> 2, j = threadIdx.x & 3;
    if (grp >= QPW) return;
    const long long q = (long long)blockIdx.x * QPW + grp;
    if (q >= nprob) return;
    double a[3][12];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < 12; ++c) a[s][c] = mats[q * 144 + 12 * (4 * s + j) + c];
    double z[12], lam;
    ep_eig12_low4_quad(a, z, lam);
#pragma unroll
    for (int i = 0; i < 12; ++i) ev[q * 48 + 12 * j + i] = z[i];
    w[q * 4 + j] = lam;
}

template <int QPW>
static void run(const double *dm, double *de, double *dw, int nprob) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((eig_kernel<QPW>), dim3((nprob + QPW - 1) / QPW), dim3(64), 0, 0, dm, de, dw, nprob);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%2d matrices per wave: %d problems in %.1f us\n", QPW, nprob, ms * 1e3);
    }
}

int main(int argc, char **argv) {
    const int nprob = argc > 1 ? atoi(argv[1]) : 8 * 1024;
    std::vector<double> h((size_t)nprob * 144);
    srand(1);
    for (int p = 0; p < nprob; ++p) {           // rank-10 Gram matrices like the 5-point M^T M; every fourth one full rank (20 rows), every 16th rank 8
        const int rows = (p % 16 == 7) ? 8 : ((p & 3) == 3 ? 20 : 10);
        double M[20][12];
        for (int i = 0; i < rows; ++i) for (double &x : M[i]) x = (rand() / (double)RAND_MAX - 0.5) * 100.0;
        for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) { double s = 0; for (int i = 0; i < rows; ++i) s += M[i][a] * M[i][b]; h[(size_t)p * 144 + 12 * a + b] = s; }
    }
    double *dm, *de, *dw;
    hipMalloc(&dm, h.size() * 8); hipMalloc(&de, (size_t)nprob * 48 * 8); hipMalloc(&dw, (size_t)nprob * 4 * 8);
    hipMemcpy(dm, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    run<16>(dm, de, dw, nprob); run<8>(dm, de, dw, nprob); run<4>(dm, de, dw, nprob); run<1>(dm, de, dw, nprob);
    run<16>(dm, de, dw, nprob / 8);
    std::vector<double> ev((size_t)nprob * 48), w((size_t)nprob * 4), ev16(ev.size());
    run<16>(dm, de, dw, nprob);
    hipMemcpy(ev16.data(), de, ev.size() * 8, hipMemcpyDeviceToHost);
    run<4>(dm, de, dw, nprob);
    hipMemcpy(ev.data(), de, ev.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(w.data(), dw, w.size() * 8, hipMemcpyDeviceToHost);
    printf("16- and 4-matrices-per-wave results %s\n", memcmp(ev.data(), ev16.data(), ev.size() * 8) ? "DIFFER" : "identical (bitwise)");
    double worst = 0;                           // residual |A v - lambda v| / |A| of every vector
    for (int p = 0; p < nprob; ++p) for (int t = 0; t < 4; ++t) {
        double r = 0, nrm = 0;
        for (int a = 0; a < 12; ++a) { double s = 0; for (int b = 0; b < 12; ++b) { s += h[(size_t)p * 144 + 12 * a + b] * ev[(size_t)p * 48 + 12 * t + b]; nrm = fmax(nrm, fabs(h[(size_t)p * 144 + 12 * a + b])); }
            r = fmax(r, fabs(s - w[(size_t)p * 4 + t] * ev[(size_t)p * 48 + 12 * t + a])); }
        worst = fmax(worst, r / nrm);
    }
    printf("worst residual |A v - lambda v| / |A| over %d vectors: %.3e\n", 4 * nprob, worst);
    if (argc > 2) {
        FILE *f = fopen(argv[2], "wb");
        const int np = nprob < 2048 ? nprob : 2048;
        fwrite(&np, 4, 1, f); fwrite(h.data(), 8, (size_t)np * 144, f); fwrite(ev.data(), 8, (size_t)np * 48, f); fwrite(w.data(), 8, (size_t)np * 4, f);
        fclose(f);
        printf("wrote %d problems to %s\n", np, argv[2]);
    }
    return 0;
}

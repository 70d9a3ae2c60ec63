// Where do single-wave workgroups land?  N workgroups x 64 threads, LDS bytes per workgroup as given: every wave records its HW_ID.
// Question: when a launch has at most one wave per SIMD of the chip (N <= 1024), does every wave get a SIMD of its own?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wave_placement_1wave.hip -o tools/ubench/wave_placement_1wave && tools/ubench/wave_placement_1wave [N] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
__global__ void __launch_bounds__(64, 1) k(unsigned *out, int spin) {
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = fma(a, 1.0000001, 1e-9);       // keep the wave alive so that all workgroups are resident together
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    if (a == 12345.0) smem[threadIdx.x] = 1;
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024, lds = argc > 2 ? atoi(argv[2]) : 1344;
    unsigned *d; hipMalloc(&d, B * 2 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL(k, dim3(B), dim3(64), lds, 0, d, 40000); hipDeviceSynchronize();
    std::vector<unsigned> h(B * 2); hipMemcpy(h.data(), d, B * 8, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    std::map<unsigned, int> per_simd, per_cu;
    for (int b = 0; b < B; ++b) {
        const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, se = (hw >> 13) & 7, sh = (hw >> 12) & 1;
        const unsigned key = (xcc << 16) | (se << 8) | (sh << 7) | (cu << 2);
        per_cu[key]++; per_simd[key | simd]++;
    }
    std::map<int, int> hs, hc;
    for (auto &kv : per_simd) hs[kv.second]++;
    for (auto &kv : per_cu) hc[kv.second]++;
    printf("%d single-wave workgroups, %d B of LDS each: %zu CUs and %zu SIMDs used\n", B, lds, per_cu.size(), per_simd.size());
    for (auto &kv : hc) printf("  %d CUs hold %d waves\n", kv.second, kv.first);
    for (auto &kv : hs) printf("  %d SIMDs hold %d waves\n", kv.second, kv.first);
    return 0;
}

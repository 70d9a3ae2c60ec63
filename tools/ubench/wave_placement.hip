// Where do the waves of a workgroup land?  1024 workgroups x 256 threads with ~31 KB of LDS each (the config-2 launch shape):
// every wave records its HW_ID (SIMD, CU, SE, XCC).  Questions: do the wave-0s of the workgroups that share a CU share a SIMD?
// which blockIdx values share a CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256, 4) k(unsigned *out, int spin) {
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = fma(a, 1.0000001, 1e-9);       // keep the wave alive so that all workgroups are resident together
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    if (a == 12345.0) smem[threadIdx.x] = 1;
}
int main() {
    const int B = 1024;
    unsigned *d; hipMalloc(&d, B * 4 * 2 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024);
    hipLaunchKernelGGL(k, dim3(B), dim3(256), 31 * 1024, 0, d, 20000); hipDeviceSynchronize();
    std::vector<unsigned> h(B * 8); hipMemcpy(h.data(), d, B * 32, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    std::map<unsigned, std::vector<int>> cu_blocks;
    int same_simd_all = 0, hist[4][4] = {};
    for (int b = 0; b < B; ++b) {
        unsigned key = 0;
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
            unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, se = (hw >> 13) & 7, sh = (hw >> 12) & 1;
            hist[w][simd]++;
            if (w == 0) key = (xcc << 16) | (se << 8) | (sh << 7) | cu;
        }
        cu_blocks[key].push_back(b);
    }
    printf("wave index -> SIMD histogram (rows: wave 0..3, cols: SIMD 0..3)\n");
    for (int w = 0; w < 4; ++w) printf("  wave %d: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("distinct CUs used: %zu\n", cu_blocks.size());
    int shown = 0; std::map<int, int> per_cu;
    for (auto &kv : cu_blocks) {
        per_cu[(int)kv.second.size()]++;
        if (shown < 6) { printf("  CU %06x: blocks", kv.first); for (int b : kv.second) { unsigned hw0 = h[(b * 4) * 2]; printf(" %d(w0 simd %u)", b, (hw0 >> 4) & 3); } printf("\n"); ++shown; }
    }
    for (auto &kv : per_cu) printf("  %d CUs hold %d workgroups\n", kv.second, kv.first);
    // leaders sharing a SIMD: per CU, how many of its workgroups have wave 0 on the same SIMD?
    std::map<int, int> worst;
    for (auto &kv : cu_blocks) { int c[4] = {}; for (int b : kv.second) c[(h[(b * 4) * 2] >> 4) & 3]++; int m = 0; for (int i = 0; i < 4; ++i) m = c[i] > m ? c[i] : m; worst[m]++; }
    for (auto &kv : worst) printf("  %d CUs: max wave-0s on one SIMD = %d\n", kv.second, kv.first);
    return 0;
}

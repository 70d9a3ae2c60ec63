// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths the PnP kernel uses (MI355X_MICROARCH.md, HBM: "FETCH_SIZE
// reports 1/2 of the bytes of a wide (16 B/lane) coalesced read; other widths are uncalibrated: calibrate on a known byte count").
// Two streaming reads of the same 1 GiB buffer (4x the Infinity Cache): read4 = one dword per lane (what load_records issues for
// channel-planar rows), read16 = one dwordx4 per lane.  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare the
// counter with the byte count printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read4(const float *p, size_t n, float *out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456f) out[0] = s;
}
__global__ void read16(const float4 *p, size_t n, float *out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    float *d, *o; hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(read4, dim3(4096), dim3(256), 0, 0, d, bytes / 4, o);
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const float4 *)d, bytes / 16, o);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: %zu (%.1f KiB)\n", bytes, bytes / 1024.0);
    return 0;
}

// Microbenchmark (development aid): the four-lane register-resident 12x12 eigen-solver of the EPnP experiment ALONE in a kernel (its own
// register budget: a[3][12] + d, e, hh and temporaries), one quad per matrix, matrices from global memory, eigenvector blocks written
// back.  Question: what does the eigen-solve cost when nothing else is live around it (inside the full EPnP kernel it took 361 us per
// batch of 30 x 1024 problems with 512 registers and AGPR copies)?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/eig12_quad.hip -o tools/ubench/eig12_quad && tools/ubench/eig12_quad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define MR_EXACT _Pragma("clang fp contract(off)")
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_bcast(double v, int j) {
    switch (j & 3) { case 0: return dpp_mov<0x00>(v); case 1: return dpp_mov<0x55>(v); case 2: return dpp_mov<0xAA>(v); default: return dpp_mov<0xFF>(v); }
}
__device__ __forceinline__ double quad_sum(double p) {
    MR_EXACT
    const double t = p + dpp_mov<0xB1>(p);            // (p0 + p1), (p2 + p3)
    return t + dpp_mov<0x4E>(t);                      // (p0 + p1) + (p2 + p3) in every lane of the quad
}

__device__ __forceinline__ void ep_eig12_quad(double (&a)[3][12], int (&col)[4], double *d, double *e) {
    MR_EXACT
    constexpr int n = 12;
    const int j = (int)threadIdx.x & 3;
    double hh[n];
#pragma unroll
    for (int i = 0; i < n; ++i) { d[i] = 0.0; e[i] = 0.0; hh[i] = 0.0; }
#pragma unroll
    for (int k = 0; k + 2 < n; ++k) {
        double pr = 0.0, pt = 0.0;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int r = 4 * s + j;
            const double xx = a[s][k] * a[s][k];
            if (r > k) pr += xx;
            if (r > k + 1) pt += xx;
        }
        const double sigma = quad_sum(pr), tail = quad_sum(pt);
        const double x0 = quad_bcast(a[(k + 1) >> 2][k], k + 1);
        if (tail == 0.0) {
            e[k] = x0; hh[k] = 0.0;
#pragma unroll
            for (int s = 0; s < 3; ++s) if (4 * s + j > k) a[s][k] = 0.0;
        } else {
            const double nrm = sqrt(sigma), alpha = x0 > 0.0 ? -nrm : nrm;
            const double h = sigma - x0 * alpha;
            if (j == ((k + 1) & 3)) a[(k + 1) >> 2][k] = x0 - alpha;
            double v[n], q[3], qf[n];
#pragma unroll
            for (int c = 0; c < n; ++c) v[c] = (c > k) ? quad_bcast(a[c >> 2][k], c) : 0.0;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                double sm = 0.0;
#pragma unroll
                for (int c = 0; c < n; ++c) if (c > k) sm += a[s][c] * v[c];
                q[s] = sm / h;
            }
            double pk = 0.0;
#pragma unroll
            for (int s = 0; s < 3; ++s) if (4 * s + j > k) pk += a[s][k] * q[s];
            const double kc = quad_sum(pk) / (h + h);
#pragma unroll
            for (int s = 0; s < 3; ++s) q[s] = q[s] - kc * a[s][k];
#pragma unroll
            for (int c = 0; c < n; ++c) qf[c] = (c > k) ? quad_bcast(q[c >> 2], c) : 0.0;
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (4 * s + j > k) {
#pragma unroll
                    for (int c = 0; c < n; ++c) if (c > k) a[s][c] = (a[s][c] - a[s][k] * qf[c]) - q[s] * v[c];
                }
            e[k] = alpha; hh[k] = h;
        }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) d[i] = quad_bcast(a[i >> 2][i], i);
    e[n - 2] = quad_bcast(a[2][n - 2], 3); e[n - 1] = 0.0;
    // orthogonal factor, backwards, in the same registers
    if (j == 2) { a[2][10] = 1.0; a[2][11] = 0.0; }       // row 10
    if (j == 3) { a[2][10] = 0.0; a[2][11] = 1.0; }       // row 11
#pragma unroll
    for (int k = n - 3; k >= 0; --k) {
        double vo[3];                                     // the stored reflector, own rows (column k is outside the block being rebuilt)
#pragma unroll
        for (int s = 0; s < 3; ++s) vo[s] = a[s][k];
#pragma unroll
        for (int c = 0; c < n; ++c) {
            if (c > k + 1) {
                if (j == ((k + 1) & 3)) a[(k + 1) >> 2][c] = 0.0;                 // row k+1
            }
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {                     // column k+1: 1 on the diagonal, 0 below
            const int r = 4 * s + j;
            if (r == k + 1) a[s][k + 1] = 1.0;
            if (r > k + 1) a[s][k + 1] = 0.0;
        }
        const double h = hh[k];
        if (h != 0.0) {
#pragma unroll
            for (int c = 0; c < n; ++c) {
                if (c > k) {
                    double pk = 0.0;
#pragma unroll
                    for (int s = 0; s < 3; ++s) if (4 * s + j > k) pk += vo[s] * a[s][c];
                    const double wc = quad_sum(pk) / h;
#pragma unroll
                    for (int s = 0; s < 3; ++s) if (4 * s + j > k) a[s][c] = a[s][c] - vo[s] * wc;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) a[s][0] = (4 * s + j == 0) ? 1.0 : 0.0;
    if (j == 0) {
#pragma unroll
        for (int c = 1; c < n; ++c) a[0][c] = 0.0;
    }
    // implicit QL with Wilkinson shifts; d / e are indexed statically inside the unrolled rotation loop, dynamically (select chains)
    // where l and m enter
    auto get = [&](const double *x, int idx) { return x[idx]; };
    for (int l = 0; l < n; ++l) {
        for (int sweep = 0; sweep < 60; ++sweep) {
            int m = n - 1;
#pragma unroll
            for (int i = n - 2; i >= 0; --i) if (i >= l && (fabs(e[i]) <= 2.220446049250313e-16 * (fabs(d[i]) + fabs(d[i + 1])))) m = i;
            if (m == l) break;
            const double dl = get(d, l), dl1 = get(d, l + 1), el = get(e, l), dm = get(d, m);
            double g = (dl1 - dl) / (2.0 * el);
            double r = sqrt(g * g + 1.0);
            g = dm - dl + el / (g + (g >= 0.0 ? r : -r));
            double sn = 1.0, cs = 1.0, pp = 0.0;
            bool underflow = false;
#pragma unroll
            for (int i = n - 2; i >= 0; --i) {
                if (i < m && i >= l && !underflow) {
                    const double f = sn * e[i];
                    const double b = cs * e[i];
                    r = sqrt(f * f + g * g);
                    e[i + 1] = r;
                    if (r == 0.0) {
                        d[i + 1] -= pp;
                        e[m] = 0.0;
                        underflow = true;
                    } else {
                        sn = f / r; cs = g / r;
                        g = d[i + 1] - pp;
                        r = (d[i] - g) * sn + 2.0 * cs * b;
                        pp = sn * r;
                        d[i + 1] = g + pp;
                        g = cs * r - b;
#pragma unroll
                        for (int s = 0; s < 3; ++s) { const double f1 = a[s][i + 1], ai = a[s][i]; a[s][i + 1] = sn * ai + cs * f1; a[s][i] = cs * ai - sn * f1; }
                    }
                }
            }
            if (underflow) continue;
            d[l] = d[l] - pp; e[l] = g; e[m] = 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) {
        int pos = 0;
#pragma unroll
        for (int t = 0; t < n; ++t) pos += ((d[t] > d[i]) || (d[t] == d[i] && t < i)) ? 1 : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) col[t] = (pos == n - 1 - t) ? i : col[t];
    }
}


template <int BOUND>
__global__ void __launch_bounds__(256, BOUND) eig_kernel(const double *mats, double *ev, int nprob) {
    const int q = (blockIdx.x * 256 + threadIdx.x) >> 2, j = threadIdx.x & 3;
    if (q >= nprob) return;
    double a[3][12];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < 12; ++c) a[s][c] = mats[(size_t)q * 144 + 12 * (4 * s + j) + c];
    int col[4] = { 0, 0, 0, 0 };
    __shared__ double sde[64][24];
    ep_eig12_quad(a, col, &sde[threadIdx.x >> 2][0], &sde[threadIdx.x >> 2][12]);
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            double v = a[s][0];
#pragma unroll
            for (int c = 1; c < 12; ++c) v = (col[t] == c) ? a[s][c] : v;
            ev[(size_t)q * 48 + 4 * (4 * s + j) + t] = v;
        }
}
int main() {
    const int nprob = 30 * 1024;
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
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bound = 1; bound <= 2; ++bound) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (bound == 1) hipLaunchKernelGGL(eig_kernel<1>, dim3((nprob * 4 + 255) / 256), dim3(256), 0, 0, dm, de, nprob);
            else hipLaunchKernelGGL(eig_kernel<2>, dim3((nprob * 4 + 255) / 256), dim3(256), 0, 0, dm, de, nprob);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("launch_bounds(256,%d): %d problems (x4 lanes) in %.1f us\n", bound, nprob, ms * 1e3);
        }
    }
    std::vector<double> ev((size_t)nprob * 48);
    hipMemcpy(ev.data(), de, ev.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0;                           // residual of the smallest eigenvector of a few problems: |A v| / |A|
    for (int p = 0; p < 64; ++p) {
        double r = 0, nrm = 0;
        for (int a = 0; a < 12; ++a) { double s = 0; for (int b = 0; b < 12; ++b) { s += h[(size_t)p * 144 + 12 * a + b] * ev[(size_t)p * 48 + 4 * b]; nrm = fmax(nrm, fabs(h[(size_t)p * 144 + 12 * a + b])); } r = fmax(r, fabs(s)); }
        worst = fmax(worst, r / nrm);
    }
    printf("null-vector residual |A v| / |A| (64 problems): %.2e\n", worst);
    return 0;
}

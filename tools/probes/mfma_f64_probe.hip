// Probe: does v_mfma_f64_16x16x4_f64 accumulate as the sequential chain
//   d = fma(a[3],b[3], fma(a[2],b[2], fma(a[1],b[1], fma(a[0],b[0], c)))) ?
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probes/mfma_f64_probe.hip -o tools/probes/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, const double* C, double* D, int ksteps)
{
    // A: ksteps x (16 x 4) as [s][i][k]; B: [s][k][j]; C, D: 16x16
    const int l = threadIdx.x;
    d4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    for (int s = 0; s < ksteps; ++s) {
        const double a = A[s * 64 + (l & 15) * 4 + (l >> 4)];
        const double b = B[s * 64 + (l >> 4) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

int main()
{
    const int S = 63;
    double *hA = (double*)malloc(S * 64 * 8), *hB = (double*)malloc(S * 64 * 8), hC[256], hD[256];
    srand(7);
    long long bad[6] = {0, 0, 0, 0, 0, 0}, total = 0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, S * 64 * 8); hipMalloc(&dB, S * 64 * 8); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    for (int trial = 0; trial < 200; ++trial) {
        for (int i = 0; i < S * 64; ++i) { hA[i] = (rand() / (double)RAND_MAX - 0.5) * 4; hB[i] = (rand() / (double)RAND_MAX - 0.5) * 4; }
        for (int i = 0; i < 256; ++i) hC[i] = trial < 100 ? 0.0 : (rand() / (double)RAND_MAX - 0.5);
        hipMemcpy(dA, hA, S * 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, hB, S * 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, 2048, hipMemcpyHostToDevice);
        const int ks = trial % 2 ? S : 1;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, ks);
        hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double h[6];
                for (int hyp = 0; hyp < 6; ++hyp) h[hyp] = hC[i * 16 + j];
                for (int s = 0; s < ks; ++s) {
                    const double* a = hA + s * 64 + i * 4;       // a[k]
                    double b[4];
                    for (int k = 0; k < 4; ++k) b[k] = hB[s * 64 + k * 16 + j];
                    for (int k = 0; k < 4; ++k) h[0] = fma(a[k], b[k], h[0]);                 // forward fma chain
                    for (int k = 3; k >= 0; --k) h[1] = fma(a[k], b[k], h[1]);                // backward chain
                    for (int k = 0; k < 4; ++k) h[2] = h[2] + a[k] * b[k];                    // unfused forward
                    { long double t = 0; for (int k = 0; k < 4; ++k) t += (long double)a[k] * b[k]; h[3] = (double)((long double)h[3] + t); }
                    { double t = fma(a[0], b[0], 0.0); t = fma(a[1], b[1], t); double u = fma(a[2], b[2], 0.0); u = fma(a[3], b[3], u); h[4] = h[4] + (t + u); }
                    { double t = 0; for (int k = 0; k < 4; ++k) t = fma(a[k], b[k], t); h[5] = h[5] + t; }
                }
                for (int hyp = 0; hyp < 6; ++hyp) bad[hyp] += (h[hyp] != hD[i * 16 + j]);
                ++total;
            }
    }
    printf("elements=%lld mismatches: fwd_fma_chain=%lld bwd_fma_chain=%lld unfused_fwd=%lld longdouble_dot=%lld pair_tree=%lld dot_then_add=%lld\n",
           total, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
    return 0;
}

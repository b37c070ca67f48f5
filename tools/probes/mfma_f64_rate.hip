// Probe: issue cost of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on gfx950 (cycles per wave instruction).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f64_rate.hip -o tools/probes/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_chain(double* out, unsigned long long* cyc, int iters)
{
    d4 acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = (d4){0, 0, 0, 0};
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&cyc[2], t0); atomicMax(&cyc[3], t1); }
}
template <int NACC>
__global__ void fma_chain(double* out, unsigned long long* cyc, int iters)
{
    double acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = q;
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = fma(a, b, acc[q]);
    }
    unsigned long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&cyc[2], t0); atomicMax(&cyc[3], t1); }
}
template <int NACC>
__global__ void add_chain(double* out, unsigned long long* cyc, int iters)
{
    double acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = q;
    double a = threadIdx.x * 0.001;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = acc[q] + a;
    }
    unsigned long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&cyc[2], t0); atomicMax(&cyc[3], t1); }
}
__global__ void divsqrt_chain(double* out, unsigned long long* cyc, int iters)
{
    double x = 1.0 + threadIdx.x * 0.01, y = 3.0 + threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) { x = x / y + 1.5; }
    unsigned long long t1 = clock64();
    for (int i = 0; i < iters; ++i) { y = sqrt(y) + 2.0; }
    unsigned long long t2 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}

__global__ void tick_calib(unsigned long long* cyc)
{
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    double x = threadIdx.x;
    for (int i = 0; i < 200000; ++i) x = fma(x, 1.0000001, 0.5);
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (x == 12345.0) cyc[0] = 0;
    if (threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
}

#define RUN(name, kern, threads, per_iter)                                                          \
    do {                                                                                            \
        unsigned long long init[4] = {0, 0, ~0ull, 0}, h[4];                                        \
        hipMemcpy(cyc, init, 32, hipMemcpyHostToDevice);                                            \
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, iters);                    \
        hipDeviceSynchronize();                                                                     \
        hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);                                               \
        printf("%-44s %8.1f ticks per wave instruction, %6.2f per SIMD\n", name, (double)(h[3] - h[2]) / iters / (per_iter), \
               (double)(h[3] - h[2]) / iters / (per_iter) / (((threads) + 255) / 256));                \
    } while (0)

int main()
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8 * 256 * 1024); hipMalloc(&cyc, 32);
    const int iters = 400000;   // long enough for the clocks to ramp; clock64 is a fixed 2.4 GHz time base
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(mfma_chain<4>, dim3(256), dim3(256), 0, 0, out, cyc, 100000);
    hipDeviceSynchronize();
    {
        hipLaunchKernelGGL(tick_calib, dim3(1), dim3(64), 0, 0, cyc);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        int wc_khz = 0; hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
        int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
        printf("clock64 ticks=%llu wall_clock64 ticks=%llu (wall clock rate %d kHz, device clock rate %d kHz) -> clock64 runs at %.1f MHz\n",
               h[0], h[1], wc_khz, clk_khz, (double)h[0] / (double)h[1] * (double)wc_khz / 1e3);
    }
    RUN("mfma_f64_16x16x4, 1 wave, dependent chain", mfma_chain<1>, 64, 1);
    RUN("mfma_f64_16x16x4, 1 wave, 2 accumulators", mfma_chain<2>, 64, 2);
    RUN("mfma_f64_16x16x4, 1 wave, 4 accumulators", mfma_chain<4>, 64, 4);
    RUN("mfma_f64_16x16x4, 4 waves (1/SIMD), 4 acc", mfma_chain<4>, 256, 4);
    RUN("mfma_f64_16x16x4, 8 waves (2/SIMD), 4 acc", mfma_chain<4>, 512, 4);
    RUN("mfma_f64_16x16x4, 16 waves (4/SIMD), 4 acc", mfma_chain<4>, 1024, 4);
    RUN("v_fma_f64, 1 wave, dependent chain", fma_chain<1>, 64, 1);
    RUN("v_fma_f64, 1 wave, 8 accumulators", fma_chain<8>, 64, 8);
    RUN("v_fma_f64, 8 waves (2/SIMD), 8 acc", fma_chain<8>, 512, 8);
    RUN("v_fma_f64, 16 waves (4/SIMD), 8 acc", fma_chain<8>, 1024, 8);
    RUN("v_add_f64, 1 wave, dependent chain", add_chain<1>, 64, 1);
    RUN("v_add_f64, 1 wave, 8 accumulators", add_chain<8>, 64, 8);
    hipLaunchKernelGGL(divsqrt_chain, dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("f64 division (dependent): %.1f cycles; f64 sqrt (dependent): %.1f cycles\n", (double)h[0] / iters, (double)h[1] / iters);
    return 0;
}

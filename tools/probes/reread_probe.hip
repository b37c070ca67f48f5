// reread_probe.hip -- where does the SECOND fetch of a window come from?
//
// The fused EEG kernel (csrc/rips.hip: eeg_window_kernel, streaming form) reads every 94,000-byte window twice: pass A
// for the channel means, pass B for the centred products.  The second fetch misses the XCD's 4 MB L2 (128 windows =
// 12 MB are in flight there) and FETCH_SIZE counts it (181 KB per window against 95 KB algorithmic), but that counter
// sits at the L2's fabric port and does not tell the 256 MiB Infinity Cache from HBM.  This probe does, by TIME: the
// same grid, the same access pattern (256 threads, 8 bytes per lane, 512 contiguous bytes per wave), four workgroups per
// CU, over a 5.2 GB buffer (nothing survives between launches):
//   mode 0  every workgroup reads ITS block twice            (the kernel's pattern)
//   mode 1  every workgroup reads its block and a DISTANT one (same bytes requested, all of them first touches)
//   mode 2  every workgroup reads its block once             (half the bytes)
// If re-reads came from HBM, mode 0 would take as long as mode 1; if they are served on the die it lies near mode 2.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/reread_probe tools/probes/reread_probe.hip && tools/probes/reread_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define BLOCK_DOUBLES 11750          // 47 x 250

__global__ void __launch_bounds__(256, 4) reread(const double* __restrict__ buf, long long n_blocks, int mode, double* __restrict__ out)
{
    __shared__ double pad[4864];     // 38 KB: the LDS footprint of the fused kernel, so that four workgroups share a CU
    const long long w = blockIdx.x;
    const double* a = buf + w * BLOCK_DOUBLES;
    const double* b = mode == 1 ? buf + ((w + n_blocks / 2) % n_blocks) * BLOCK_DOUBLES : a;
    double s = 0.0;
    for (int i = threadIdx.x; i < BLOCK_DOUBLES; i += 256) s += a[i];
    pad[threadIdx.x] = s;
    __syncthreads();                 // (the second pass starts when the first is complete, as in the kernel)
    double t = pad[(threadIdx.x + 1) & 255];
    if (mode != 2)
        for (int i = threadIdx.x; i < BLOCK_DOUBLES; i += 256) t += b[i] * 1.0000001;
    if (t == 123.456) out[w] = t;    // (never true: keeps the loads)
}

int main()
{
    const long long n_blocks = 55224;                       // one band batch of the features leg
    const size_t bytes = (size_t)n_blocks * BLOCK_DOUBLES * sizeof(double);
    double *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, n_blocks * sizeof(double)) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"own block twice   ", "own + distant     ", "own block once    "};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(reread, dim3((unsigned)n_blocks), dim3(256), 0, 0, buf, n_blocks, mode, out);   // warm-up of the code, not of the data
            hipEventRecord(e0, 0);
            for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(reread, dim3((unsigned)n_blocks), dim3(256), 0, 0, buf, n_blocks, mode, out);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            const double req = (mode == 2 ? 1.0 : 2.0) * bytes;
            if (rep == 1)
                printf("mode %d  %s %8.3f ms   %7.1f GB/s requested   (%.2f GB requested per launch)\n", mode, names[mode], ms, req / ms / 1e6, req / 1e9);
        }
    return 0;
}

// Which SIMD does wave w of a workgroup run on?  (The Rips kernels have stretches in which ONE wave of the workgroup
// works and the others wait at a barrier: if that wave is wave 0 everywhere and wave 0 always sits on the same SIMD, the
// solo stretches of all the workgroups resident on a CU queue up on one SIMD while three stand idle.)
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/probes/wave_simd.hip -o tools/probes/wave_simd && tools/probes/wave_simd
// Every wave leaves HW_REG_HW_ID (gfx9 layout: wave 3:0, SIMD 5:4, pipe 7:6, CU 11:8, SH 12, SE 15:13) and spins a
// little so that the grid fills the chip the way the real kernels do (LDS per workgroup as theirs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>

template <int NT>
__global__ void __launch_bounds__(NT) k_ids(uint32_t* out, int spin)
{
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID, all 32 bits
    uint32_t x = threadIdx.x;
    for (int i = 0; i < spin; ++i) { x = x * 1664525u + 1013904223u; smem[(x >> 8) % 1024] = (unsigned char)x; }
    __syncthreads();
    if (lane == 0) out[blockIdx.x * (NT / 64) + wave] = hw;
    if (x == 0x12345u) out[0] = smem[5];
}

template <int NT>
static void run(const char* name, int lds, int n_wg)
{
    constexpr int NW = NT / 64;
    uint32_t* d; hipMalloc(&d, (size_t)n_wg * NW * 4);
    hipFuncSetAttribute((const void*)k_ids<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k_ids<NT><<<n_wg, NT, lds>>>(d, 20000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h((size_t)n_wg * NW);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    long simd0[4] = {0, 0, 0, 0};
    std::map<int, long> rel[8];
    for (int g = 0; g < n_wg; ++g) {
        const int s0 = (h[(size_t)g * NW] >> 4) & 3;
        ++simd0[s0];
        for (int w = 0; w < NW; ++w) ++rel[w][(int)(((h[(size_t)g * NW + w] >> 4) & 3) - s0 + 4) & 3];
    }
    printf("%s: %d workgroups of %d threads, %d B of LDS\n  SIMD of wave 0:", name, n_wg, NT, lds);
    for (int s = 0; s < 4; ++s) printf("  SIMD%d %5.1f %%", s, 100.0 * simd0[s] / n_wg);
    printf("\n  SIMD of wave w relative to wave 0 (most frequent, share):");
    for (int w = 0; w < NW; ++w) {
        int best = 0; long cnt = -1;
        for (auto& kv : rel[w]) if (kv.second > cnt) { cnt = kv.second; best = kv.first; }
        printf("  w%d:+%d (%.0f %%)", w, best, 100.0 * cnt / n_wg);
    }
    printf("\n");
    hipFree(d);
}

int main()
{
    run<256>("EEG-like", 38 * 1024, 256 * 4 * 8);
    run<512>("audio-like", 53760, 256 * 3 * 8);
    run<512>("wide audio", 79 * 1024, 256 * 2 * 8);
    return 0;
}

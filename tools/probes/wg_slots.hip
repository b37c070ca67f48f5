// How many one-wave workgroups does a CU hold at once?  (The Wasserstein, filter and tau kernels are one wave per
// workgroup: a limit of 16 workgroups per CU would halve their residency whatever LDS and registers allow.)
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/probes/wg_slots.hip -o tools/probes/wg_slots && tools/probes/wg_slots
// Every workgroup spins for a fixed time; a grid of 256 CUs x k workgroups takes one spin as long as k fits, two beyond.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT>
__global__ void __launch_bounds__(NT) k_spin(unsigned long long ticks, unsigned* sink)
{
    const unsigned long long t0 = wall_clock64();
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0x12345u) *sink = x;
}
template <int NT>
static void sweep(const char* name)
{
    unsigned* sink; hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("%s: ms for 256 CUs x k workgroups spinning 0.2 ms each:", name);
    for (int k : {8, 16, 17, 24, 32, 33, 40, 48, 64, 65}) {
        k_spin<NT><<<256 * k, NT>>>(20000ull, sink);            // 100 MHz ticks
        hipDeviceSynchronize();
        hipEventRecord(a);
        k_spin<NT><<<256 * k, NT>>>(20000ull, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("  k=%d %.2f", k, ms);
    }
    printf("\n");
    hipFree(sink);
}
int main() { sweep<64>("one wave per workgroup"); sweep<128>("two waves per workgroup"); sweep<256>("four waves per workgroup"); return 0; }

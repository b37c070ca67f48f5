// Issue cost of single vector instructions on gfx950, in SIMD cycles per wave64 instruction, at 1 / 2 / 4 / 6 waves per SIMD.
// What DESIGN.md's roofline_valu and the instruction diet of rips.hip are priced with.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/probes/valu_rate.hip -o tools/probes/valu_rate && tools/probes/valu_rate
// Every wave runs ITER x 32 copies of one instruction on 8 independent register sets (no dependent issue inside a
// set of eight); workgroups of 256 threads = one wave per SIMD, k workgroups per CU via the grid (256 CUs x k).
// cycles = elapsed x shader clock / (instructions per wave x waves per SIMD); the clock is measured under load
// (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY32(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL32(NAME, ASM)                                                                    \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                    \
    uint32_t a[8], b = seed | 1u, c = threadIdx.x;                                             \
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                                 \
    for (int it = 0; it < ITER; ++it) {                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                        \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20"); \
        }                                                                                      \
    }                                                                                          \
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                     \
    if (s == 0x12345u) out[threadIdx.x] = s;                                                   \
}

#define KERNEL64(NAME, ASM)                                                                    \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                    \
    uint64_t a[8], b = ((uint64_t)seed << 20) | 3u; uint32_t c = (threadIdx.x & 7) + 1;        \
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                                 \
    for (int it = 0; it < ITER; ++it) {                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                        \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20"); \
        }                                                                                      \
    }                                                                                          \
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                     \
    if (s == 0x12345u) out[threadIdx.x] = (uint32_t)s;                                         \
}

#define KERNELF64(NAME, ASM)                                                                   \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                    \
    double a[8], b = 1.0 + seed * 1e-9, c = 1e-9 * threadIdx.x;                                \
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + i + threadIdx.x;                                  \
    for (int it = 0; it < ITER; ++it) {                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                        \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20"); \
        }                                                                                      \
    }                                                                                          \
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                       \
    if (s == 12345.0) out[threadIdx.x] = 1;                                                    \
}

#define KERNELF32(NAME, ASM)                                                                   \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                    \
    float a[8], b = 1.0f + seed * 1e-6f, c = 1e-6f * threadIdx.x;                              \
    for (int i = 0; i < 8; ++i) a[i] = 1.0f + i + threadIdx.x;                                 \
    for (int it = 0; it < ITER; ++it) {                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                        \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20"); \
        }                                                                                      \
    }                                                                                          \
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                        \
    if (s == 12345.0f) out[threadIdx.x] = 1;                                                   \
}

KERNEL32(k_add_u32,      "v_add_u32 %0, %0, %1")
KERNEL32(k_and_b32,      "v_and_b32 %0, %0, %1")
KERNEL32(k_lshl_b32,     "v_lshlrev_b32 %0, 1, %0")
KERNEL32(k_lshl_add,     "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL32(k_add3,         "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_mul_lo,       "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_u24,      "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u24,      "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_bcnt,         "v_bcnt_u32_b32 %0, %0, %1")
KERNEL32(k_ffbl,         "v_ffbl_b32 %0, %0")
KERNEL32(k_min_u32,      "v_min_u32 %0, %0, %1")
KERNEL32(k_perm,         "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_bfe,          "v_bfe_u32 %0, %0, 3, 9")
KERNEL32(k_mbcnt,        "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL32(k_cmp_cnd,      "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_dpp_mov,      "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_dpp_add,      "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_readlane,     "v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0")
KERNEL32(k_sad_u8,       "v_sad_u8 %0, %0, %1, %2")
KERNEL64(k_lshl_b64,     "v_lshlrev_b64 %0, 1, %0")
KERNEL64(k_lshl_b64_v,   "v_lshlrev_b64 %0, %2, %0")
KERNEL64(k_lshr_b64_v,   "v_lshrrev_b64 %0, %2, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL64(k_mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %2, %2, %0")
KERNEL64(k_cmp_u64,      "v_cmp_lt_u64 vcc, %0, %1")
KERNEL64(k_mov_b64,      "v_mov_b64 %0, %1")
KERNELF64(k_add_f64,     "v_add_f64 %0, %0, %1")
KERNELF64(k_mul_f64,     "v_mul_f64 %0, %0, %1")
KERNELF64(k_fma_f64,     "v_fma_f64 %0, %0, %1, %2")
KERNELF64(k_sqrt_f64,    "v_sqrt_f64 %0, %0")
KERNELF64(k_rcp_f64,     "v_rcp_f64 %0, %0")
KERNELF32(k_fma_f32,     "v_fma_f32 %0, %0, %1, %2")
KERNELF32(k_pk_fma_f32,  "v_fmac_f32 %0, %1, %2")
KERNELF32(k_sqrt_f32,    "v_sqrt_f32 %0, %0")
KERNELF32(k_rcp_f32,     "v_rcp_f32 %0, %0")
KERNELF32(k_cvt_f64_f32, "v_cvt_u32_f32 %0, %0")

// LDS: one ds_read_b32 per lane, conflict-free, results consumed by a waitcnt at the end of each group of eight
__global__ void __launch_bounds__(256) k_ds_read(uint32_t* out, uint32_t seed) {
    __shared__ uint32_t buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = i + seed;
    __syncthreads();
    uint32_t addr = (threadIdx.x & 255) * 4, a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(addr));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 0x12345u) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_ds_read_b64(uint32_t* out, uint32_t seed) {
    __shared__ uint64_t buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = i + seed;
    __syncthreads();
    uint32_t addr = (threadIdx.x & 255) * 8; uint64_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(addr));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 0x12345u) out[threadIdx.x] = (uint32_t)s;
}
// scalar side: s_add_u32 chain beside nothing (scalar issue cost per wave)
__global__ void __launch_bounds__(256) k_salu(uint32_t* out, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_readfirstlane(seed + i);
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(a[i]) : "s"(seed) : "scc");
        }
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 0x12345u) out[threadIdx.x] = s;
}

// shader clock under load: s_memtime (shader cycles) against s_memrealtime (100 MHz) around the v_add_u32 loop
__global__ void __launch_bounds__(256) k_clock(uint32_t* out, uint32_t seed, unsigned long long* clk) {
    uint32_t a[8], b = seed | 1u;
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 0x12345u) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Op { const char* name; kern_t k; int per; };      // per: instructions per asm statement

int main() {
    int dev = 0; hipSetDevice(dev);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    int cus = p.multiProcessorCount;
    uint32_t* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<Op> ops = {
        {"v_add_u32", k_add_u32, 1}, {"v_and_b32", k_and_b32, 1}, {"v_lshlrev_b32", k_lshl_b32, 1}, {"v_lshl_add_u32", k_lshl_add, 1},
        {"v_add3_u32", k_add3, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_u32_u24", k_mul_u24, 1}, {"v_mad_u32_u24", k_mad_u24, 1},
        {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_ffbl_b32", k_ffbl, 1}, {"v_min_u32", k_min_u32, 1}, {"v_perm_b32", k_perm, 1},
        {"v_bfe_u32", k_bfe, 1}, {"v_mbcnt_lo", k_mbcnt, 1}, {"v_cmp_lt_u32+v_cndmask", k_cmp_cnd, 2},
        {"v_mov_b32_dpp row_shr", k_dpp_mov, 1}, {"v_add_u32_dpp row_shr", k_dpp_add, 1}, {"v_readlane+v_add(sgpr)", k_readlane, 2},
        {"v_sad_u8", k_sad_u8, 1},
        {"v_lshlrev_b64 (imm)", k_lshl_b64, 1}, {"v_lshlrev_b64 (vgpr)", k_lshl_b64_v, 1}, {"v_lshrrev_b64 (vgpr)", k_lshr_b64_v, 1},
        {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_cmp_lt_u64", k_cmp_u64, 1},
        {"v_mov_b64", k_mov_b64, 1},
        {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_sqrt_f64", k_sqrt_f64, 1},
        {"v_rcp_f64", k_rcp_f64, 1},
        {"v_fma_f32", k_fma_f32, 1}, {"v_fmac_f32", k_pk_fma_f32, 1}, {"v_sqrt_f32", k_sqrt_f32, 1}, {"v_rcp_f32", k_rcp_f32, 1},
        {"v_cvt_u32_f32", k_cvt_f64_f32, 1},
        {"ds_read_b32 (no conflicts)", k_ds_read, 1}, {"ds_read_b64 (no conflicts)", k_ds_read_b64, 1}, {"s_add_u32", k_salu, 1},
    };
    auto run = [&](kern_t k, int wps) {
        k<<<cus * wps, 256>>>(out, 7);                   // warm
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k<<<cus * wps, 256>>>(out, 7);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    unsigned long long* clk; hipMalloc(&clk, 16);
    unsigned long long hclk[2];
    k_clock<<<cus * 6, 256>>>(out, 7, clk);
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    double ghz = (double)hclk[0] / (double)hclk[1] * 0.1;
    printf("CUs %d; s_memtime / s_memrealtime (100 MHz) under a full v_add_u32 load: %.3f GHz; s_memtime cycles per v_add_u32 at 6 waves/SIMD: %.2f\n",
           cus, ghz, (double)hclk[0] / ((double)ITER * 32 * 6));
    if (ghz < 0.5 || ghz > 3.0) { ghz = 2.4; printf("  (counter ratio implausible: pricing with the nominal 2.4 GHz)\n"); }
    printf("%-30s %8s %8s %8s %8s   cycles per wave64 instruction and SIMD at 1 / 2 / 4 / 6 waves per SIMD\n", "instruction", "1", "2", "4", "6");
    for (auto& o : ops) {
        printf("%-30s", o.name);
        for (int wps : {1, 2, 4, 6}) {
            float ms = run(o.k, wps);
            double cyc = ms * 1e6 * ghz / ((double)ITER * 32 * o.per * wps);
            printf(" %8.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}

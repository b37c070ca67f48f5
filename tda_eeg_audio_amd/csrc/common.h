// common.h -- shared host/device helpers of libtdaeeg (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/tdaeeg.h"

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint16_t u16;

struct tda_ctx {
    int device = 0;
    int words_dm = 2;       // H1 class capacity (x64) for distance-matrix input
    int words_cloud = 1;    // ... for point clouds
    int retry_policy = 0;   // TDA_RETRY_*
    unsigned long long* retry_ctr = nullptr;   // tda_set_retry_counter: device u64[4]
    unsigned long long* total_scratch = nullptr;   // class vectors of the last rung of the Rips ladders (rips.hip: TOT_SLOTS x 8.3 MB)
    int h1_order = 0;       // TDA_ORDER_*
    // Lists of the windows a widening pass has to redo (rips.hip: retry_collect): one buffer per STREAM -- the Rips calls
    // of a stream, eager or replayed from a HIP graph captured on it, run one after the other, so a stream's list is never
    // in use twice, while calls on different streams never share one.  TDA_RETRY_SLOTS buffers, allocated with the context
    // and assigned to streams in the order they are first seen (more streams than buffers: the last one is shared and
    // retry_shared counts it).  [0] = entries, [1] = workgroups done, the window indices from [4] on.  A call with more
    // windows than retry_cap replaces all buffers (the old ones stay alive for graphs that hold their addresses).
    int* retry_buf[32] = {};
    hipStream_t retry_stream[32] = {};
    int retry_streams = 0;
    int retry_shared = 0;
    int retry_cap = 0;
    std::vector<void*> retired;
    // host-API staging workspace (grown on demand, only by the host-pointer twins)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    std::string err;
    // one-shot kernel probes (tda_set_kernel_probe), one slot per probed kernel
    struct Probe { hipEvent_t start = nullptr, stop = nullptr; unsigned long long* span = nullptr; bool armed = false; };
    Probe probe[4];
};

// brackets ONE kernel launch with the armed probe of `which` (if any) and disarms it
struct ProbeScope {
    tda_ctx* ctx; hipStream_t st; int which; bool on; unsigned long long* span;
    ProbeScope(tda_ctx* c, int w, hipStream_t s)
        : ctx(c), st(s), which(w), on(w > 0 && w < 4 && c->probe[w].armed), span(nullptr)
    {
        if (on) {
            span = ctx->probe[which].span;
            if (ctx->probe[which].start) (void)hipEventRecord(ctx->probe[which].start, st);
        }
    }
    ~ProbeScope()
    {
        if (on) {
            if (ctx->probe[which].stop) (void)hipEventRecord(ctx->probe[which].stop, st);
            ctx->probe[which] = tda_ctx::Probe();
        }
    }
};

#define TDA_HIP(ctx, call)                                                          \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) {                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);        \
            return TDA_ERR_HIP;                                                     \
        }                                                                           \
    } while (0)

#define TDA_FAIL(ctx, code, msg)                                                    \
    do { (ctx)->err = (msg); return (code); } while (0)

// ---- wave64 cross-lane helpers (lane index must be wave-uniform) ----
__device__ __forceinline__ u32 rl32(u32 v, int lane) { return (u32)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ u64 rl64(u64 v, int lane)
{
    u32 lo = rl32((u32)v, lane), hi = rl32((u32)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 uni64(u64 v)
{
    u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
    u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
// sqrt of a double >= 0, bit for bit what sqrt() returns: the library's own sequence (v_rsq_f64 and three fused
// corrections) without its rescaling of tiny arguments and its class test.  Arguments outside {0} u [2^-767, inf) --
// squares of coordinates below 1e-115, infinities, NaNs -- take the library's path, decided per wave on the high word.
__device__ __forceinline__ double sqrt_rn(double x)
{
    const bool zero = x == 0.0;
    const u32 hi = (u32)(__double_as_longlong(x) >> 32);
    if (__builtin_expect(__ballot(hi - 0x10000000u > 0x6fefffffu && !zero) != 0ull, 0)) return sqrt(x);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return zero ? x : g;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- wave64 reductions on the DPP network (a few VALU cycles per step; __shfl_xor goes through
// ds_bpermute and costs an LDS round trip per step).  Result is wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double uni_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
#define TDA_DPP_REDUCE_F64(v, OP)                                                     \
    do {                                                                              \
        double o__;                                                                   \
        o__ = dpp_f64<0xB1, 0xF>(v); v = OP(o__, v);   /* quad_perm [1,0,3,2] */      \
        o__ = dpp_f64<0x4E, 0xF>(v); v = OP(o__, v);   /* quad_perm [2,3,0,1] */      \
        o__ = dpp_f64<0x141, 0xF>(v); v = OP(o__, v);  /* row_half_mirror     */      \
        o__ = dpp_f64<0x140, 0xF>(v); v = OP(o__, v);  /* row_mirror          */      \
        o__ = dpp_f64<0x142, 0xA>(v); v = OP(o__, v);  /* row_bcast:15        */      \
        o__ = dpp_f64<0x143, 0xC>(v); v = OP(o__, v);  /* row_bcast:31        */      \
    } while (0)
#define TDA_MIN_(a, b) ((a) < (b) ? (a) : (b))
#define TDA_MAX_(a, b) ((a) > (b) ? (a) : (b))
#define TDA_ADD_(a, b) ((a) + (b))
__device__ __forceinline__ double wave_min_f64_dpp(double v) { TDA_DPP_REDUCE_F64(v, TDA_MIN_); return uni_f64(v, 63); }
__device__ __forceinline__ double wave_max_f64_dpp(double v) { TDA_DPP_REDUCE_F64(v, TDA_MAX_); return uni_f64(v, 63); }

// The DPP modifier sits on the min / max instruction itself (one instruction per step; the builtin route costs a
// v_mov_b32_dpp plus the operation).  s_nop 1 = the two wait states between a VALU write and a DPP read of a VGPR.
#define TDA_DPP_STEPS_(OP)                                                                   \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"        \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"        \
    "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"            \
    "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                 \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"               \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"               \
    "s_nop 1"
__device__ __forceinline__ int wave_max_i32_dpp(int v)
{
    asm volatile(TDA_DPP_STEPS_("v_max_i32_dpp") : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ u32 wave_min_u32_dpp(u32 x)
{
    int v = (int)x;
    asm volatile(TDA_DPP_STEPS_("v_min_u32_dpp") : "+v"(v));
    return (u32)__builtin_amdgcn_readlane(v, 63);
}

// order-preserving float32 <-> uint32 (handles negative values; NaN sorts last)
__device__ __forceinline__ u32 f32_sortable(float f)
{
    u32 b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float sortable_f32(u32 s)
{
    u32 b = (s & 0x80000000u) ? (s & 0x7fffffffu) : ~s;
    return __uint_as_float(b);
}

// launch-side entry points implemented in the .hip files
size_t rips_total_scratch_bytes();
#define TDA_RETRY_SLOTS 32
tda_status retry_lists_reserve(tda_ctx*, int n_win);      // (rips.hip) all slots to at least n_win entries
tda_status launch_corr_dist(tda_ctx*, const double*, int, int, int, double*, double*, hipStream_t);
tda_status launch_corr_dist_sliding(tda_ctx*, const double*, int, int, int, int, double*, double*, int*, hipStream_t);
tda_status launch_corr_to_dist(tda_ctx*, const double*, int, int, int, double*, hipStream_t);
tda_status launch_rips_dm(tda_ctx*, const double*, int, int, double, int, double*, int, int*, double*, int, int*,
                          int*, hipStream_t);
tda_status launch_eeg_windows(tda_ctx*, const double*, int, int, int, double, double*, double*, double*, int, int*, double*,
                              int, int*, int*, hipStream_t);
tda_status launch_eeg_sliding(tda_ctx*, const double*, int, int, int, int, int, const int*, int, double, double*, double*,
                              double*, int, int*, double*, int, int*, int*, int*, hipStream_t);
tda_status launch_rips_cloud(tda_ctx*, const double* win_or_pc, const int* tau_or_npts, int n_win, int n_t_or_pcap,
                             int dim, int subsample, int mode, int normalise, double thresh, double*, int, int*,
                             double*, int, int*, int* n_points, int*, hipStream_t);
tda_status launch_sosfiltfilt(tda_ctx*, const double*, int, int, const double*, const double*, int, int, double*, double*,
                              hipStream_t, int n_filt = 1);
tda_status launch_filtfilt(tda_ctx*, const double*, int, int, const double*, const double*, const double*, int, int, double*,
                           double*, hipStream_t, int n_filt = 1);
tda_status launch_upfirdn(tda_ctx*, const double*, long long, const double*, int, int, int, long long, long long, double*,
                          hipStream_t);
tda_status launch_hilbert_env(tda_ctx*, const double*, int, const double*, double*, hipStream_t);
tda_status launch_tau(tda_ctx*, const double*, int, int, int, int*, hipStream_t);
tda_status launch_tau_segments(tda_ctx*, const double*, const int*, int, int, int, int*, int*, hipStream_t);
tda_status launch_recording_rows(tda_ctx*, const double*, const double*, const int*, const double*, const double*, const int*,
                                 int, double*, const int*, const int*, int*, hipStream_t);
tda_status launch_features(tda_ctx*, const double*, const int*, int, int, double*, hipStream_t);
tda_status launch_diagram_finish(tda_ctx*, const tda_diagram_set*, int, int, hipStream_t);
tda_status launch_aggregate(tda_ctx*, const double*, const double*, const int*, int, double*, hipStream_t);
tda_status launch_nanmean(tda_ctx*, const double*, const int*, int, double*, hipStream_t);
tda_status launch_spearman(tda_ctx*, const double*, const double*, int, const int*, int, const int*, int, double*, hipStream_t);
tda_status launch_wasserstein(tda_ctx*, const double*, const int*, int, const double*, const int*, int, const int*,
                              const int*, int, double*, int*, hipStream_t);

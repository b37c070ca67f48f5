// rips.hip -- batched Vietoris-Rips H0/H1 persistence for gfx950 (MI355X), wave64.
//
// Replaces ripser(dm, maxdim=1, thresh, distance_matrix=True) at scripts/utils.py:140 /
// scripts/tda_eeg_classification_v2.py:170-175 and ripser(pc_norm, maxdim=1, thresh) at
// scripts/utils.py:131 (incl. the Takens embedding utils.py:107-116 and the min-max
// normalisation utils.py:127-130 in front of it).
//
// One window per workgroup (256 threads for distance matrices, 512 for point clouds); everything
// between the input read and the diagram rows lives in LDS / registers.  The algorithm is NOT
// ripser's (heap columns do not map to a GPU); it is an edge-parallel formulation of the same
// persistence pairing:
//
//  P0  all n(n-1)/2 float32 edge lengths as sortable 32-bit keys; edges longer than
//      min(thresh, enclosing radius) are dropped (they cannot contribute a row);
//  P1  the rank of every edge in the order of (length, a, b) WITHOUT sorting: a monotone linear map spreads the lengths of
//      the window over NB >= E buckets (about one edge per bucket); rank = #edges in earlier buckets + #members of the
//      own bucket that precede the edge (count, DPP scan, scatter, walk: see rank_edges / rank_edges_narrow);
//  P2  rank[(a,b)] = r for the r-th edge (triangular u16 table, 0x7fff for edges beyond the effective
//      threshold) and ord[r] = (a,b);
//  P3  sweep over the filtration in chunks of NT consecutive edges, ONE EDGE PER LANE:
//      a. common-neighbour mask M_r = { v : (a,v) and (b,v) older than r }: the adjacency bit rows of
//         the chunk start give the bulk; a vertex can only have become common since then through an
//         edge of this chunk, and those few candidates (bit rows of the chunk's own edges) are
//         confirmed with two rank look-ups -- no sequential adjacency state inside a chunk;
//      b. edges with M_r = 0 are the only candidates for negative (spanning-forest) edges: wave 0
//         walks them in rank order with the component labels in registers and leaves one bit each,
//                                      merge -> H0 death at |e|   /   else -> a new H1 class is born;
//         the H0 rows, the class bits of the births and their table entries are then allocated by
//         prefix sums over ballots and written by the lanes that own the edges;
//      c. every other edge is killed at once by a triangle (a,b,v*), v* preferably a common
//         neighbour that predates the chunk:  psi[e] = psi[a,v*] ^ psi[b,v*], where psi[edge] in LDS
//         is the class of the cycle "edge + forest path" as a bit vector over the alive H1 classes;
//      d. LINK ARGUMENT: common neighbours v, v' that are adjacent span a tetrahedron with (a,b)
//         whose other two faces entered earlier, so triangles (a,b,v) and (a,b,v') carry the same
//         boundary class.  Only common neighbours outside the connected component of v* in the link
//         (closure over adjacency bit rows) are tested: psi[e]^psi[a,v]^psi[b,v].  The earliest
//         non-zero one in the chunk kills the YOUNGEST class in it (elder rule): that class is
//         substituted out of the whole psi table and the pair (birth, |e|) is emitted.  This happens
//         exactly once per H1 class that ever dies (tens per window).  All non-trivial triangles of a
//         chunk are listed at once; wave 0 reduces the list in registers (the substitutions are
//         linear maps) and the psi table is rewritten once per chunk through a class -> image table.
//      Chunks in which no class is alive and none is born skip c and d (every psi entry is zero); once, in that
//      state, every remaining edge already has a common neighbour, nothing can happen any more and the sweep
//      stops.  Candidate-rich chunks decide b by Boruvka rounds of the whole workgroup.
//      The multiset of (birth,death) pairs equals that of any persistence algorithm on the
//      same filtration; tie order inside equal diameters does not change it.
//  P4  rows are written as float64 (float32-exact) pairs: H0 ascending death then the essential
//      rows; H1 rows are ordered by a second tiny kernel (descending birth).
//
// LDS per workgroup: 26 KB (n = 47, 64 classes), 32 KB (128 classes) .. 52.5 KB (point cloud of up to 124 points, first
// pass: a third of a CU) .. 79.2 KB (the wide passes of a 124-point cloud).
// Class capacity ladder: a first pass, then widening passes that redo only flagged windows (tda_set_retry_policy), then a
// last rung with the class vectors in HBM and no capacity limit (rips_sweep_total).
// No MFMA: this is irregular integer work; the roofline that binds it is vector issue / LDS latency.
#include "common.h"
#include "corr_dist_dev.h"
#include <stdlib.h>

#ifndef CLOUD_NB
#define CLOUD_NB 8192          // ranking buckets of the point-cloud flavour
#endif
#ifndef NARROW_NT
#define NARROW_NT 512          // workgroup of the narrow first pass (384 works too: measured slower)
#endif
#ifndef NARROW_WAVES
#define NARROW_WAVES 6         // launch bound, waves per SIMD: three workgroups of 512 per CU = 24 waves = 80 VGPRs
#endif
#if defined(TDA_PROFILE) && !defined(TDA_PROFILE_STOPS_ONLY)
#define NARROW_NB (NARROW_NT == 384 ? 1536 : 1024)   // (the phase counters take 384 bytes of LDS: fewer buckets in this build)
#else
#define NARROW_NB (NARROW_NT == 384 ? 1536 : 2048)   // ... and its ranking buckets (a multiple of 2 NT that fits)
#endif
#define NARROW_EMAX 7626       // ... and the edges it is built for: 124 points (a Takens cloud of a 250-sample window, tau = 1)
#define NT_MAX 512            // largest workgroup (point-cloud flavour); distance-matrix flavour uses 256
#define RANK_NONE 0x7fffu

// ---- optional phase profiling (make PROFILE=1): cycle sums per phase over all windows ----
#ifdef TDA_PROFILE
// Per-workgroup sums live in LDS (thread 0 adds, no atomics) and are flushed to the global table once per
// window: hundreds of workgroups hammering the same global counters at every mark cost more than the phases.
__device__ unsigned long long g_prof[48];
#ifdef TDA_PROFILE_STOPS_ONLY
// (make PROFILE=1 EXTRA=-DTDA_PROFILE_STOPS_ONLY: only the stop points, for exact instruction counts per phase under
// rocprofv3 -- no timers, no static LDS, so the layouts and the residency are the product's)
#define PROF_BEGIN() do {} while (0)
#define PROF_RESUME() do {} while (0)
#define PROF_MARK(i) do {} while (0)
#define PROF_COUNT(i, v) do {} while (0)
#define PROF_FLUSH() do {} while (0)
#else
__shared__ unsigned long long prof_lds[48];
#define PROF_BEGIN()                                                       \
    if (threadIdx.x < 48) prof_lds[threadIdx.x] = 0ull;                    \
    __syncthreads();                                                       \
    unsigned long long prof_t0 = clock64()
#define PROF_RESUME() unsigned long long prof_t0 = clock64()
#define PROF_MARK(i)                                                       \
    do {                                                                   \
        unsigned long long prof_t1 = clock64();                            \
        if (threadIdx.x == 0) prof_lds[i] += prof_t1 - prof_t0;            \
        prof_t0 = prof_t1;                                                 \
    } while (0)
#define PROF_COUNT(i, v) do { if (threadIdx.x == 0) prof_lds[i] += (unsigned long long)(v); } while (0)
#define PROF_FLUSH()                                                       \
    do {                                                                   \
        __syncthreads();                                                   \
        if (threadIdx.x < 48 && prof_lds[threadIdx.x]) atomicAdd(&g_prof[threadIdx.x], prof_lds[threadIdx.x]); \
    } while (0)
#endif
// experiment knob of the diagnostic build: the Rips kernels return after phase n (1: keys, 2: ranking); 0 = run all
__device__ int g_stop_after;
extern "C" __attribute__((visibility("default"))) int tda_profile_stop_after(int n)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stop_after), &n, sizeof(int)) != hipSuccess;
}
#define PROF_STOP(n, cleanup) do { if (g_stop_after == (n)) { cleanup; return; } } while (0)
// inside the sweep: n + 100 * chunk stops after phase n of that chunk (11..19: phases, 20: end of the chunk)
#define PROF_STOPC(n, cleanup) do { if (g_stop_after == (n) + 100 * prof_chunk) { cleanup; return; } } while (0)
extern "C" __attribute__((visibility("default"))) int tda_profile_read(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 48) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[48] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#else
#define PROF_BEGIN() do {} while (0)
#define PROF_RESUME() do {} while (0)
#define PROF_MARK(i) do {} while (0)
#define PROF_COUNT(i, v) do {} while (0)
#define PROF_FLUSH() do {} while (0)
#define PROF_STOP(n, cleanup) do {} while (0)
#define PROF_STOPC(n, cleanup) do {} while (0)
#endif

// Order between LDS accesses of ONE wave is kept by the hardware (the LDS serves a wave's instructions in issue order): a
// flag written after data needs no s_waitcnt in between, only that the compiler keeps the two in program order.
#define LDS_ORDER() asm volatile("" ::: "memory")

#define LAUNDER(v) asm volatile("" : "+v"(v))
#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

struct SweepShared {
    u64 alive[8];
    int k0, merges, status, clen, k1, nk, more;
};

typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
typedef short pk_i16 __attribute__((ext_vector_type(2)));

template <int W, typename WT>
struct __attribute__((aligned(sizeof(WT)))) Psi {
    WT w[W];            // W words of WT (u32: 32 classes per word, u64: 64)
};
template <int W, typename WT>
__device__ __forceinline__ Psi<W, WT> pxor(Psi<W, WT> a, Psi<W, WT> b)
{
    Psi<W, WT> r;
#pragma unroll
    for (int i = 0; i < W; ++i) r.w[i] = a.w[i] ^ b.w[i];
    return r;
}
template <int W, typename WT>
__device__ __forceinline__ bool pnz(Psi<W, WT> a)
{
    WT o = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) o |= a.w[i];
    return o != 0;
}
template <int W, typename WT>
__device__ __forceinline__ Psi<W, WT> pzero()
{
    Psi<W, WT> r;
#pragma unroll
    for (int i = 0; i < W; ++i) r.w[i] = 0;
    return r;
}

__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

__device__ __forceinline__ int tri2(int v) { return (v * (v - 1)) >> 1; }
// index of unordered pair {x,y}, x != y
__device__ __forceinline__ int pair_index(int x, int y) { return x > y ? tri2(x) + y : tri2(y) + x; }

// flat edge index e = tri2(a)+b (a > b)  ->  a
__device__ __forceinline__ int edge_row(int e)
{
    // e < 2^13: 1 + 8e is exact in float and the hardware's v_sqrt_f32 (one ulp, no denormal / class fix-ups: the
    // correctly rounded sqrtf of -fno-fast-math costs 18 instructions more, tools/probes/valu_rate.hip) puts the estimate
    // within 1e-4 of (1 + sqrt(1 + 8e)) / 2, i.e. off by at most one row -- one branch-free correction
    const int a = (int)((1.0f + __builtin_amdgcn_sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);
    const int t = tri2(a);
    return a + (t + a <= e ? 1 : 0) - (t > e ? 1 : 0);
}

// (a << 8 | b) of every flat edge index, as a constant table in device memory: 17 KB that every CU keeps in its vector
// L1, read with one coalesced load per wave where edge_row() spends 20 vector instructions (the loops over all edges:
// keys and ord).  Padded by one stride of the widest workgroup so that a loop may fetch one trip ahead unguarded.
#ifndef TDA_NO_EDGE_TABLE
struct EdgeTable {
    u16 v[8192 + 512];
    constexpr EdgeTable() : v()
    {
        int a = 1, b = 0;
        for (int e = 0; e < 8192 + 512; ++e) { v[e] = (u16)((a << 8) | b); if (++b == a) { ++a; b = 0; } }
    }
};
__device__ const EdgeTable g_edge_ab = EdgeTable();
__device__ __forceinline__ u32 edge_ab(int e) { return (u32)g_edge_ab.v[e]; }
#else
__device__ __forceinline__ u32 edge_ab(int e) { const int a = edge_row(e); return (u32)((a << 8) | (e - tri2(a))); }
#endif

// ---------------------------------------------------------------------------------
// Workgroup-wide OR / AND of a predicate with ONE barrier (the library's __syncthreads_or / _and take three
// and an LDS atomic).  Every wave leaves the verdict of its ballot in its slot, one barrier, everybody reads the
// slots.  Two sets of slots are used alternately: a wave writes set s again only two votes later, after it has
// passed the barrier of the vote in between -- which no wave reaches before it has read set s.  All threads of
// the workgroup must take part in every vote (uniform control flow), as with __syncthreads.
// ---------------------------------------------------------------------------------
struct WgVote {
    u32* slots;      // 16 u32 in LDS, 16-byte aligned
    int parity;
};
template <int NT>
__device__ __forceinline__ bool wg_any(WgVote& v, bool p)
{
    static_assert(NT % 64 == 0 && NT <= 512, "slots are read as one or two 16-byte words; slots beyond NT / 64 hold zero");
    const bool w = __ballot(p) != 0ull;
    u32* s = v.slots + 8 * v.parity;
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = w ? 1u : 0u;
    __syncthreads();
    v.parity ^= 1;
    const uint4 x = *reinterpret_cast<const uint4*>(s);
    u32 r = x.x | x.y | x.z | x.w;
    if (NT > 256) { const uint4 y = *reinterpret_cast<const uint4*>(s + 4); r |= y.x | y.y | y.z | y.w; }
    return r != 0u;
}
template <int NT>
__device__ __forceinline__ bool wg_all(WgVote& v, bool p) { return !wg_any<NT>(v, !p); }

// ---------------------------------------------------------------------------------
// P1 + P2: the rank of every edge in the order of (key, a, b), WITHOUT sorting.
// The float32 lengths of one window are spread over (0, effective threshold]; a monotone linear map sends every edge
// to one of NB >= E buckets (about one edge per bucket), so that
//   rank(e) = #edges in earlier buckets + #edges of its own bucket that precede it in (key, flat index) order
// (the flat index tri2(a)+b orders edges of equal length as (a,b) does).  Four sweeps over the edges:
//   count    one LDS atomic per edge on a packed 16-bit counter (two buckets per word; counts < 65,536 never carry)
//   scan     exclusive prefix over the buckets: every thread owns NB/NT consecutive buckets, wave scan on the DPP network
//   scatter  members[slot] = e, slot from a returning atomic on the bucket's cursor (the order inside a bucket is
//            arbitrary and never used)
//   rank     every edge walks the few members of its bucket; rank[e] = r, ord[r] = (a << 8 | b)
// ~60 vector instructions and ~15 LDS accesses per edge, against ~300 / ~56 for the seven-pass LSD radix sort this
// replaces (and 10 barriers instead of 28).  Tie-heavy input degrades gracefully: a bucket of s equal lengths costs
// s^2 comparisons -- bounded by E^2 / NT per thread when ALL lengths are equal.
// Edges longer than the effective threshold take no part (rank RANK_NONE).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan_i32(int v)
{
#define TDA_DPP_ADD_(CTRL, RM) v += __builtin_amdgcn_update_dpp(0, v, CTRL, RM, 0xF, true)
    TDA_DPP_ADD_(0x111, 0xF);     // row_shr:1
    TDA_DPP_ADD_(0x112, 0xF);     // row_shr:2
    TDA_DPP_ADD_(0x114, 0xF);     // row_shr:4
    TDA_DPP_ADD_(0x118, 0xF);     // row_shr:8
    TDA_DPP_ADD_(0x142, 0xA);     // row_bcast:15 -> rows 1, 3
    TDA_DPP_ADD_(0x143, 0xC);     // row_bcast:31 -> rows 2, 3
#undef TDA_DPP_ADD_
    return v;
}

__device__ __forceinline__ int edge_flat(u32 pk) { return tri2((int)(pk >> 8)) + (int)(pk & 255u); }

// monotone map from a sortable key in [kmin, teff] to a bucket in [0, NB)
struct Bucketer {
    float dmin, scale, top;
    __device__ __forceinline__ int operator()(u32 key) const
    {
        // fminf returns the other operand for a NaN (0 * inf when all lengths are equal or the threshold is infinite)
        const float x = fminf((sortable_f32(key) - dmin) * scale, top);
        const int b = (int)x;
        return b < 0 ? 0 : b;
    }
};

// Effective threshold = min(thresh, enclosing radius).  At the enclosing radius
// r_enc = min_v max_u d(v,u) some vertex is adjacent to every other one, the complex is a cone and
// stays one: H0 is connected, every H1 class is dead and every later edge is killed at once by the
// apex with zero persistence.  Edges longer than r_enc therefore cannot contribute a diagram row
// (ripser applies the same cut when no threshold is given); dropping them shortens the sweep.
// vmax[v] (u32 sortable keys, LDS) must hold max_u key(v,u) on entry.  teff_out / kmin_out: the effective threshold and
// the smallest key of the window (kmin_thread: the minimum over the keys this thread produced).  The number of edges
// within the threshold falls out of the ranking (rank_edges returns it).  red: u32[4] scratch in LDS.
template <int NT>
__device__ __forceinline__ void effective_threshold(int n, u32 tkey, const u32* vmax, u32 kmin_thread, u32* red, u32& teff_out, u32& kmin_out)
{
    const int tid = threadIdx.x;
    // enclosing radius: every wave takes the minimum over the (<= 128) vertices, two per lane, on the DPP network
    const int lane = tid & 63;
    const u32 m0 = lane < n ? vmax[lane] : 0xffffffffu, m1 = lane + 64 < n ? vmax[lane + 64] : 0xffffffffu;
    const u32 renc = wave_min_u32_dpp(m0 < m1 ? m0 : m1);
    teff_out = renc < tkey ? renc : tkey;
    // smallest key of the window: every thread brings the minimum over the keys it produced
    if (tid == 0) red[1] = 0xffffffffu;
    __syncthreads();
    const u32 kmin = wave_min_u32_dpp(kmin_thread);
    if (lane == 63) atomicMin(&red[1], kmin);
    __syncthreads();
    kmin_out = red[1];
}

// key32: E keys (flat index order).  members: E u16 (may share its LDS with ord).  cursor: NB u16 (as NB/2 packed
// words).  wsum: NT/64 ints.
template <int NT, int NB, bool WANT_KEYS>
__device__ __forceinline__ int rank_edges(const u32* key32, int E, u32 teff, u32 kmin, u16* members, u32* cursor, int* wsum, u16* rank,
                          u16* ord, u32* skey)
{
    static_assert(NB % (2 * NT) == 0, "every thread scans whole words");
    constexpr int WPT = NB / NT / 2;                  // packed words per thread in the scan
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Bucketer bk;
    bk.dmin = sortable_f32(kmin);
    bk.top = (float)(NB - 1);
    bk.scale = bk.top / (sortable_f32(teff) - bk.dmin);
    for (int i = tid; i < NB / 2; i += NT) cursor[i] = 0u;
    __syncthreads();
    // ---- count (the bucket of every edge waits in its slot of the rank table: computed once) ----
#pragma unroll 4
    for (int e = tid; e < E; e += NT) {
        const u32 k = key32[e];
        u32 b = 0xffffu;
        if (k <= teff) { b = (u32)bk(k); atomicAdd(&cursor[b >> 1], 1u << (16 * (b & 1))); }
        rank[e] = (u16)b;
    }
    __syncthreads();
    // ---- scan ----
    {
        u32 w[WPT];
        int s = 0;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const u32 x = cursor[tid * WPT + i];
            const int c0 = (int)(x & 0xffffu), c1 = (int)(x >> 16);
            w[i] = (u32)s | ((u32)(s + c0) << 16);    // exclusive prefix inside the thread
            s += c0 + c1;
        }
        const int incl = wave_incl_scan_i32(s);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int basep = incl - s;
        for (int q = 0; q < wave; ++q) basep += wsum[q];
        const u32 add = (u32)basep * 0x00010001u;     // both halves (sums stay below 65,536)
#pragma unroll
        for (int i = 0; i < WPT; ++i) cursor[tid * WPT + i] = w[i] + add;
    }
    __syncthreads();
    // ---- scatter: afterwards cursor[b] = end of bucket b = start of bucket b + 1 ----
#pragma unroll 4
    for (int e = tid; e < E; e += NT) {
        const u32 b = rank[e];
        if (b != 0xffffu) {
            const int sh = 16 * (int)(b & 1u);
            const u32 old = atomicAdd(&cursor[b >> 1], 1u << sh);
            members[(old >> sh) & 0xffffu] = (u16)e;
        }
    }
    __syncthreads();
    // ---- rank ----
    const u16* cur16 = reinterpret_cast<const u16*>(cursor);
    const int Ev = uni((int)cur16[NB - 1]);           // end of the last bucket = edges within the effective threshold
#pragma unroll 4
    for (int e = tid; e < E; e += NT) {
        const u32 b = rank[e];
        u32 r = RANK_NONE;
        if (b != 0xffffu) {
            const u32 k = key32[e];
            const int lo = b ? (int)cur16[b - 1] : 0, hi = (int)cur16[b];
            int c = 0;
            for (int j = lo; j < hi; ++j) {
                const u32 m = members[j];
                const u32 km = key32[m];
                c += (km < k || (km == k && m < (u32)e)) ? 1 : 0;
            }
            r = (u32)(lo + c);
        }
        rank[e] = (u16)r;
    }
    __syncthreads();                                  // ord may overlay `members`: every read of it is done
    // ---- ord (and the sorted keys) from the rank table ----
#pragma unroll 4
    for (int e = tid; e < E; e += NT) {
        const u32 r = rank[e];
        if (r != RANK_NONE) {
            ord[r] = (u16)edge_ab(e);
            if (WANT_KEYS) skey[r] = key32[e];
        }
    }
    __syncthreads();
    return Ev;
}

// ---------------------------------------------------------------------------------
// The narrow first pass of the point-cloud kernel: one third of a CU's LDS per workgroup, and a layout that is FIXED at
// compile time (sized for 124 points in up to three dimensions, whatever the clouds of the batch): every LDS address of
// the sweep is then a lane-dependent index plus an immediate offset of the ds instruction.  With the offsets as kernel
// arguments the compiler kept two dozen loop-invariant addresses in registers across the sweep and, at the 80 VGPRs
// that three 512-thread workgroups per CU leave, spilled them to scratch: the reloads sat in the single-wave
// stretches of every chunk.
//   ranking:  [keys 4E][members 2E][cursors 2 NB][scratch]                          ...     [points]
//   sweep:    [rank 2E][class vectors by rank ->   (one region, NL::REGION bytes)   <- ord][misc][points]
// ---------------------------------------------------------------------------------
#if defined(TDA_PROFILE) && !defined(TDA_PROFILE_STOPS_ONLY)
#define NARROW_LDS ((160 * 1024 / 3 / 1280) * 1280 - 384)     // (the phase counters take 384 bytes of static LDS)
#else
// LDS is handed out in granules of 1,280 bytes on gfx950 (160 KB / 128): a third of a CU is 42 granules = 53,760 bytes
// (54,608 = 163,840 / 3 rounded down to 16 is rounded UP to 43 granules by the hardware and only two workgroups fit:
// measured, the kernel ran at the residency of the wide layout)
#define NARROW_LDS ((160 * 1024 / 3 / 1280) * 1280)
#endif
#define NARROW_PMAX 124
#define NARROW_DIM 3
#ifdef TDA_DEBUG_PTS
#define NL_GUARD 16
#else
#define NL_GUARD 0
#endif
#define NL_A16(x) (((x) + 15) & ~15)
struct NL {                                              // byte offsets of the narrow layout
    static constexpr int E = NARROW_EMAX;
    static constexpr int RANK = 0;
    static constexpr int PSI = NL_A16(2 * E) + NL_GUARD;
    static constexpr int MEMBERS = NL_A16(4 * E);
    static constexpr int CURSOR = NL_A16(MEMBERS + 2 * E);
    static constexpr int RSCR = CURSOR + 2 * NARROW_NB;  // vmax, min-max scratch, wave sums, reductions: 720 bytes
    static constexpr int TOTAL = NARROW_LDS;
    static constexpr int AUX = TOTAL - NL_GUARD - NL_A16(NARROW_PMAX * NARROW_DIM * 8);
    static constexpr int MISC = AUX - NL_GUARD - NL_A16(8752 + 8 * 32);      // MISC_BYTES(32), checked on the host side
    static constexpr int REGION = MISC - NL_GUARD - PSI;  // class vectors from its start, ord at its end
    static_assert(RSCR + 720 <= AUX - NL_GUARD, "the ranking arrays must end in front of the points");
    static_assert(MEMBERS >= 4 * (E + 1) && CURSOR >= MEMBERS + 2 * (E + 3), "room for the walk's padding (rank_edges_narrow)");
    static_assert(REGION >= 2 * E + 8 * 512, "the first chunks must fit: class vectors of 2 NT ranks next to the whole of ord");
};

// ---------------------------------------------------------------------------------
// The same ranking for the NARROW first pass of the point-cloud kernel (three workgroups per CU: 54.6 KB of LDS per
// window instead of 79): no bucket index parked in the rank table (recomputed for the scatter), 2,048 buckets instead
// of 8,192 -- and the rank table
// takes the place of the keys: every thread keeps the ranks of its (at most 16) edges in eight registers across the
// barrier that ends the last read of a key, and `ord` is written from them at the end of the region it shares with
// the class vectors (see rips_sweep).  LDS during the ranking: keys 4E | members 2E | cursors 2 NB.
// The walk over a bucket (3.6 members on average now) fetches four members per trip: two LDS round trips per four
// members instead of two per member.
// ---------------------------------------------------------------------------------
template <int NT, int NB>
__device__ __forceinline__ int rank_edges_narrow(const u32* key32, int E, u32 teff, u32 kmin, u16* members, u32* cursor, int* wsum,
                                                 u16* rank, u16* ord)
{
    static_assert(NB % (2 * NT) == 0, "every thread scans whole words");
    static_assert(NB <= 0x8000, "bucket indices are parked in 16 bits, 0xffff = beyond the threshold");
    constexpr int WPT = NB / NT / 2;                  // packed words per thread in the scan
    constexpr int EPT = 2 * ((NARROW_EMAX + 2 * NT - 1) / (2 * NT));      // edges per thread, at most (even)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Bucketer bk;
    bk.dmin = sortable_f32(kmin);
    bk.top = (float)(NB - 1);
    bk.scale = bk.top / (sortable_f32(teff) - bk.dmin);
    for (int i = tid; i < NB / 2; i += NT) cursor[i] = 0u;
    __syncthreads();
    // ---- count; the bucket of every edge stays in a register (two per VGPR) until its rank takes the place ----
    u32 rk[EPT / 2];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NT;
        u32 b = 0xffffu;
        if (e < E) {
            const u32 k = key32[e];
            if (k <= teff) { b = (u32)bk(k); atomicAdd(&cursor[b >> 1], 1u << (16 * (b & 1))); }
        }
        if (i & 1) rk[i >> 1] |= b << 16; else rk[i >> 1] = b;
    }
    __syncthreads();
    // ---- scan ----
    {
        u32 w[WPT];
        int s = 0;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const u32 x = cursor[tid * WPT + i];
            const int c0 = (int)(x & 0xffffu), c1 = (int)(x >> 16);
            w[i] = (u32)s | ((u32)(s + c0) << 16);    // exclusive prefix inside the thread
            s += c0 + c1;
        }
        const int incl = wave_incl_scan_i32(s);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int basep = incl - s;
        for (int q = 0; q < wave; ++q) basep += wsum[q];
        const u32 add = (u32)basep * 0x00010001u;     // both halves (sums stay below 65,536)
#pragma unroll
        for (int i = 0; i < WPT; ++i) cursor[tid * WPT + i] = w[i] + add;
        // The walk below fetches four members per trip WITHOUT looking at the end of its bucket: members of later
        // buckets have larger keys and never count, and behind the last bucket lie three entries that name a key of
        // 0xffffffff (both in the padding that the fixed layout leaves behind the two arrays)
        if (tid == NT - 1) {
            const int total = basep + s;
            members[total] = (u16)NARROW_EMAX; members[total + 1] = (u16)NARROW_EMAX; members[total + 2] = (u16)NARROW_EMAX;
            const_cast<u32*>(key32)[NARROW_EMAX] = 0xffffffffu;
        }
    }
    __syncthreads();
    // ---- scatter: afterwards cursor[b] = end of bucket b = start of bucket b + 1 ----
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const u32 b = i & 1 ? rk[i >> 1] >> 16 : rk[i >> 1] & 0xffffu;
        if (b != 0xffffu) {
            const int sh = 16 * (int)(b & 1u);
            const u32 old = atomicAdd(&cursor[b >> 1], 1u << sh);
            members[(old >> sh) & 0xffffu] = (u16)(tid + i * NT);
        }
    }
    __syncthreads();
    // ---- rank: rank(e) = start of its bucket + the members that precede it in (key, flat index) order.  Four members
    // per trip; (key, index) pairs are compared as 64-bit numbers; what a trip reads beyond the end of the bucket
    // belongs to later buckets or is the padding (see the scan) and does not precede the edge ----
    const u16* cur16 = reinterpret_cast<const u16*>(cursor);
    const int Ev = uni((int)cur16[NB - 1]);           // end of the last bucket = edges within the effective threshold
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const u32 e = (u32)(tid + i * NT);
        const u32 b = i & 1 ? rk[i >> 1] >> 16 : rk[i >> 1] & 0xffffu;
        u32 r = RANK_NONE;
        if (b != 0xffffu) {
            const u32 k = key32[e];
            const u64 me = ((u64)k << 32) | e;
            const int lo = b ? (int)cur16[b - 1] : 0, hi = (int)cur16[b];
            int c = 0;
            for (int j = lo; j < hi; j += 4) {
                u32 m[4], km[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) m[t] = (u32)members[j + t];
#pragma unroll
                for (int t = 0; t < 4; ++t) km[t] = key32[m[t]];
#pragma unroll
                for (int t = 0; t < 4; ++t) c += ((((u64)km[t] << 32) | m[t]) < me) ? 1 : 0;
            }
            r = (u32)(lo + c);
        }
        if (i & 1) rk[i >> 1] = (rk[i >> 1] & 0xffffu) | (r << 16); else rk[i >> 1] = (rk[i >> 1] & 0xffff0000u) | r;
    }
    __syncthreads();                                  // rank and ord lie over the keys / members: every read of those is done
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NT;
        if (e < E) {
            const u32 r = i & 1 ? rk[i >> 1] >> 16 : rk[i >> 1] & 0xffffu;
            rank[e] = (u16)r;
            if (r != RANK_NONE) ord[r] = (u16)edge_ab(e);
        }
    }
    __syncthreads();
    return Ev;
}

struct RipsOut {
    double* h0; int h0_cap; int* h0_cnt;
    double* h1; int h1_cap; int* h1_cnt;
    int* status;
    int* retry_list = nullptr;                  // widening passes: [0] = windows to redo, their indices from [4] on (retry_collect)
};

struct RipsLayout {
    int off_members;                            // ranking phase: keys at 0, then the bucket members
    int off_rank, off_ord, off_aux, off_misc;   // sweep phase: psi at 0, rank, ord; then aux and misc
    int total;
    // narrow layout (make_narrow_layout / struct NL): rank at 0, class vectors by rank behind it, `ord` at the end of their region
    int off_psi, psi_bytes, off_cursor, off_rscr;   // (psi_bytes: the region class vectors and ord share; narrow only)
    int guard[5], n_guard;                      // guard build: where the sentinels end (regions that start there)
};

// Guard build (make DEBUG_PTS=1): 16 sentinel bytes in front of rank, ord, aux and misc and at the end of the
// layout, written when the ranking phase is over (it legitimately uses the space across the regions) and checked
// after the sweep; an overwritten sentinel sets status bit TDA_WIN_LDS_GUARD.  The product build has no gaps.
#define TDA_WIN_LDS_GUARD 0x100
// One wave works while the others of its workgroup wait at a barrier: it goes first at its SIMD's issue port
// (the other workgroups' waves there lose nothing in total, the waiting seven get going sooner).
#ifndef TDA_SOLO_PRIO
#define TDA_SOLO_PRIO 3
#endif
#define TDA_SOLO_BEGIN() __builtin_amdgcn_s_setprio(TDA_SOLO_PRIO)
#define TDA_SOLO_END() __builtin_amdgcn_s_setprio(0)

// Every poll on an LDS word gives up after this many idle trips (a trip is an LDS round trip plus a sleep: >= 150
// cycles, so >= 10 M cycles; the longest legitimate wait -- seven hand-overs, or a dependency chain of ~30 links, on a
// fully loaded CU -- stays below 1,000 trips).  A wave that gives up raises the workgroup's poison word, lets the
// others through and the window ends with TDA_WIN_NOT_CONVERGED: a defect in a dependency table becomes a status
// bit, never a hang.
#define TDA_POLL_LIMIT (1 << 16)
#ifndef TDA_TURN_SLEEP
#define TDA_TURN_SLEEP 4        // x 64 cycles between two looks of a wave that is not next (0 / 2 / 4 / 8: within 0.5 %, measured)
#endif
#ifdef TDA_DEBUG_PTS
// guard build only: tda_debug_inject(1) makes wave 3 keep the turn of phase a to itself, (2) makes one apparent edge
// of every chunk wait for itself -- tests/test_gpu_stress.py expects status 8 back, not a hang
__device__ int g_fault;
extern "C" __attribute__((visibility("default"))) int tda_debug_inject(int what)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fault), &what, sizeof(int)) != hipSuccess;
}
#define TDA_FAULT(n) (g_fault == (n))
#else
#define TDA_FAULT(n) false
#endif
#ifdef TDA_DEBUG_PTS
#define GUARD_BYTES 16
__device__ __forceinline__ void guard_write(unsigned char* smem, const RipsLayout& L)
{
    if ((int)threadIdx.x < 4 * L.n_guard) reinterpret_cast<u32*>(smem + L.guard[threadIdx.x >> 2] - GUARD_BYTES)[threadIdx.x & 3] = 0xC0FFEE00u + threadIdx.x;
    __syncthreads();
}
__device__ __forceinline__ int guard_check(unsigned char* smem, const RipsLayout& L)
{
    __syncthreads();
    bool bad = false;
    if ((int)threadIdx.x < 4 * L.n_guard) bad = reinterpret_cast<u32*>(smem + L.guard[threadIdx.x >> 2] - GUARD_BYTES)[threadIdx.x & 3] != 0xC0FFEE00u + threadIdx.x;
    return __syncthreads_or(bad ? 1 : 0) ? TDA_WIN_LDS_GUARD : 0;
}
#else
#define GUARD_BYTES 0
__device__ __forceinline__ void guard_write(unsigned char*, const RipsLayout&) {}
__device__ __forceinline__ int guard_check(unsigned char*, const RipsLayout&) { return 0; }
#endif

// misc block (byte offsets inside off_misc); the class tables at its end are sized by the variant
#define MISC_COMP 0                                          // int comp[128] / u32 vmax[128] (before the sweep)
#define MISC_CAND 512                                        // u64 cand[16]
#define MISC_WV (512 + 128)                                  // 64 B: wave sums of the sort, then the vote slots of the sweep
#define MISC_MIN (512 + 192)                                 // u32[4]: reductions; list count / earliest key
#define MISC_DONE (512 + 208)                                // u8 done[NT_MAX]
#define MISC_SORTCNT (512 + 224)                             // ranking phase only: u16 cursor[NB] (over the sweep's part)
#define MISC_SHARED (MISC_DONE + NT_MAX)                     // SweepShared (96 B)
#define MISC_CKEY (MISC_SHARED + 96)                         // float ckey[NT_MAX]: lengths of this chunk's candidate edges
#define MISC_LIST MISC_CKEY                                  // phase d reuses ckey + the tail: triangle list, then image table
#define MISC_LIST_BYTES (4 * NT_MAX + 1280)
#define MISC_ADJ (MISC_LIST + MISC_LIST_BYTES)               // u64 adj[128][2]: adjacency bit rows at chunk start
#define MISC_ADJC (MISC_ADJ + 128 * 16)                      // u64 adjc[128][2]: adjacency bit rows of the chunk's own edges
#define MISC_BRANK (MISC_ADJC + 128 * 16)                    // int brank[ncls], then float bkey[ncls]
#define MISC_BYTES(ncls) (MISC_BRANK + 8 * (ncls))

// ---------------------------------------------------------------------------------
// The sweep (phase P3).  KEYFN(r, a, b) returns the float32 length of sorted edge r = (a,b).
// All control flow is workgroup-uniform; ord/rank/psi/misc live in LDS.
// ---------------------------------------------------------------------------------
// NARROW (the first pass of the point-cloud kernel at three workgroups per CU): the class vectors are indexed by RANK
// and share ONE region of LDS with `ord`:
//   * psi[r] belongs to the edge of rank r and grows upwards from the start of the region, 4 bytes per rank; ord[r]
//     lies at the END of the region, 2 bytes per rank (ord[E - 1] in its last two bytes).  A chunk [r0, r0 + NT) reads
//     ord from r0 on only (its own edges, the end-of-story walk) and class vectors below r0 + NT only, so the two never
//     meet as long as 4 (r0 + NT) <= region - 2 (E - r0): for a 124-point cloud classes may stay alive up to rank
//     ~5,500 of 7,626 (they live to ~2,000 on average, beyond 5,100 in 0.6 % of the windows).  A busy chunk beyond
//     that flags the window TDA_WIN_CLASS_OVERFLOW for the wide pass, like a window that runs out of class bits;
//     quiet chunks need no class vectors and go on;
//   * a triangle side (flat index j) is psi[rank[j]] -- one more dependent LDS read in the triangle list, none in
//     phase c (the ranks of the two reference edges are needed there anyway); the rewrite after a kill walks the
//     vectors directly, without `ord`;
//   * every chunk clears the vectors of its own ranks first (the region held `ord` entries of dead ranks there).
// (forced inline: left to itself the compiler outlined one instantiation of the point-cloud kernel, and an outlined
// sweep gets its LDS pointers as generic 64-bit ones -- flat loads instead of ds_read, every phase 20-40 % slower)
template <int NT, int NVW, int W, typename WT, bool NARROW, class KEYFN>
__device__ __forceinline__ void rips_sweep(int n, int E, int Ev, const u16* rank, const u16* ord, Psi<W, WT>* psi, int psi_bytes,
                           unsigned char* misc, KEYFN keyfn, double* h0, int h0_cap, double* h1, int h1_cap,
                           int& out_k0, int& out_k1, int& out_status)
{
    static_assert(!NARROW || W == 1, "the narrow layout is the point-cloud first pass");
    int psi_cap = 0;                                 // NARROW: set per chunk
    // The thread index is handed to every chunk through an opaque move (LAUNDER): scaled copies of it (tid * 4, lane * 8,
    // ...: the indices of two dozen LDS arrays) are then recomputed per chunk where they are used instead of being hoisted
    // out of the chunk loop and kept -- or, at 80 VGPRs, spilled to scratch and reloaded in the single-wave stretches.
    const int tid0 = threadIdx.x;
    const int lane0 = tid0 & 63, wave = uni(tid0 >> 6);
    int* brank = reinterpret_cast<int*>(misc + MISC_BRANK);
    u64* cand = reinterpret_cast<u64*>(misc + MISC_CAND);
    unsigned char* done = misc + MISC_DONE;
    SweepShared* shared = reinterpret_cast<SweepShared*>(misc + MISC_SHARED);
    u64* adj = reinterpret_cast<u64*>(misc + MISC_ADJ);      // adj[2*v + w]: neighbours of v among vertices 64w..64w+63
    constexpr int WB = 8 * (int)sizeof(WT);          // class bits per word
    float* bkey = reinterpret_cast<float*>(misc + MISC_BRANK + 4 * WB * W);
    u64* adjc = reinterpret_cast<u64*>(misc + MISC_ADJC);    // adjc[2*v + w]: neighbours of v through edges of this chunk

    if constexpr (!NARROW) for (int e = tid0; e < E; e += NT) psi[e] = pzero<W, WT>();
    for (int i = tid0; i < WB * W; i += NT) { brank[i] = -1; bkey[i] = 0.f; }
    for (int i = tid0; i < 256; i += NT) { adj[i] = 0ull; adjc[i] = 0ull; }
    if (tid0 == 0) {
        u32* lc = reinterpret_cast<u32*>(misc + MISC_MIN);     // list header: [0] entries, [1] earliest key
        lc[0] = 0u; lc[1] = 0xffffffffu; lc[2] = 0u;               // [2]: whose turn it is in phase a
        lc[3] = 0u;                                                // [3]: poison (a poll gave up: TDA_POLL_LIMIT)
        shared->status = 0;
    }
    if (tid0 < 16) {                                                // vote slots and ballots of waves that do not exist (NT = 384)
        reinterpret_cast<u32*>(misc + MISC_WV)[tid0] = 0u;
        cand[tid0] = 0ull;
    }
    __syncthreads();

    WgVote vote{reinterpret_cast<u32*>(misc + MISC_WV), 0};      // (the sort's wave sums are done with)
    WT alive[W];
#pragma unroll
    for (int c = 0; c < W; ++c) alive[c] = 0;
    int k0 = 0, k1 = 0, merges = 0, status = 0;
    int compA = lane0, compB = lane0 + 64;     // component labels of vertices lane0 / lane0+64 (used by wave 0)

    int clen = NT;
    int ordcls = 0;                              // wave 0, one class word: lane0 i holds the i-th oldest class alive
    int cov_next = tid0;                          // coverage check: first edge of this thread's residue class not yet seen covered
#ifdef TDA_PROFILE
    int prof_rneed = 0, prof_chunk = -1;
#endif
    PROF_RESUME();
    for (int r0 = 0; r0 < Ev && !status; r0 += clen) {
        int tid = tid0;
        LAUNDER(tid);
        const int lane = tid & 63;
        clen = NT;
        PROF_MARK(15);
#ifdef TDA_PROFILE
        ++prof_chunk;
#endif
        PROF_COUNT(23, 1);
        const int r = r0 + tid;                      // (adjc was cleared at the end of the previous chunk)
        const bool valid = r < Ev;
        int a = 1, b = 0;
        if (valid) { const u32 pk = ord[r]; a = (int)(pk >> 8); b = (int)(pk & 255u); }     // ord[r] = (a << 8 | b), a > b
        const int tab = tri2(a) + b;
        const int slot = NARROW ? r : tab;               // where the class vector of this edge lives
        // NARROW: psi_cap = the ranks whose vectors lie clear of ord[r0 ...] in this chunk (psi and ord share a region)
        if constexpr (NARROW) {
            psi_cap = (psi_bytes - 2 * (E - r0)) >> 2;
            if (r < psi_cap) psi[r] = pzero<W, WT>();
        }
        // ---- a. common neighbours of (a,b) before edge r ----
        // Neighbours through edges that predate the chunk come from the adjacency bit rows `adj`.  The chunk's own
        // edges enter a second set of rows, `adjc`, WAVE BY WAVE in rank order: when wave w takes its turn, adjc holds
        // exactly the chunk's edges of the waves before it -- all of them older than every edge of wave w -- so
        // (adj | adjc)[a] & (adj | adjc)[b] is exact up to the 64 edges of the wave itself, and only vertices that
        // become common through one of THOSE are confirmed against the rank table (about one per edge).  The waves
        // hand the turn on through a word in LDS (the LDS serves the accesses of a wave in program order: rows read,
        // own bits added, turn passed).  Before, all edges of the chunk entered adjc at once and every edge confirmed
        // every vertex that the whole chunk had brought near it: 9 on average, 15-20 for the busiest lane of a wave,
        // two rank lookups each -- a third of the vector instructions of the audio kernel.
        u64 M[NVW], M0[NVW];     // M: common neighbours before edge r; M0: those already common at chunk start
        u64 Aj[NVW], Bj[NVW];
#pragma unroll
        for (int w = 0; w < NVW; ++w) { M[w] = 0; M0[w] = 0; Aj[w] = 0; Bj[w] = 0; }
        if (valid) {
#pragma unroll
            for (int w = 0; w < NVW; ++w) { Aj[w] = adj[2 * a + w]; Bj[w] = adj[2 * b + w]; M0[w] = Aj[w] & Bj[w]; }
        }
        {
            volatile u32* turn = reinterpret_cast<volatile u32*>(misc + MISC_MIN) + 2;
            if (wave > 0) {
                // A wave whose turn is several hand-overs away sleeps longer between two looks (a hand-over takes ~600 cycles;
                // fewer looks = fewer vector instructions of waiting waves -- no measurable effect on the time, though)
                int trips = 0;
                for (;;) {
                    const u32 t = (u32)uni((int)*turn);
                    if (t >= (u32)wave) break;
                    if (++trips > TDA_POLL_LIMIT) { if (lane == 0) turn[1] = 1u; break; }     // poison: see TDA_POLL_LIMIT
                    if (t + 1u < (u32)wave) __builtin_amdgcn_s_sleep(TDA_TURN_SLEEP); else __builtin_amdgcn_s_sleep(0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (valid) {
#pragma unroll
                for (int w = 0; w < NVW; ++w) M[w] = (Aj[w] | adjc[2 * a + w]) & (Bj[w] | adjc[2 * b + w]);
                atomicOr(reinterpret_cast<unsigned long long*>(&adjc[2 * a + (b >> 6)]), 1ull << (b & 63));
                atomicOr(reinterpret_cast<unsigned long long*>(&adjc[2 * b + (a >> 6)]), 1ull << (a & 63));
            }
            LDS_ORDER();
            if (lane == 0 && !(TDA_FAULT(1) && wave == 3)) *turn = (u32)wave + 1u;
        }
        if (valid) {
            // through the wave's own edges (later waves may have added theirs meanwhile: the rank test rejects them)
            const int ta0 = tri2(a), tb0 = tri2(b);
#pragma unroll
            for (int w = 0; w < NVW; ++w) {
                u64 cc = (Aj[w] | adjc[2 * a + w]) & (Bj[w] | adjc[2 * b + w]) & ~M[w];
                while (cc) {
                    const int v0 = 64 * w + __builtin_ctzll(cc); cc &= cc - 1ull;
                    const bool ok1 = cc != 0ull; const int v1 = ok1 ? 64 * w + __builtin_ctzll(cc) : v0; cc &= cc - 1ull;
                    const int t0 = tri2(v0), t1 = tri2(v1);
                    const u32 xa0 = rank[v0 < a ? ta0 + v0 : t0 + a], xb0 = rank[v0 < b ? tb0 + v0 : t0 + b];
                    const u32 xa1 = rank[v1 < a ? ta0 + v1 : t1 + a], xb1 = rank[v1 < b ? tb0 + v1 : t1 + b];
                    if ((int)(xa0 > xb0 ? xa0 : xb0) < r) M[w] |= 1ull << (v0 & 63);
                    if (ok1 && (int)(xa1 > xb1 ? xa1 : xb1) < r) M[w] |= 1ull << (v1 & 63);
                }
            }
        }
        u64 many = 0;
#pragma unroll
        for (int w = 0; w < NVW; ++w) many |= M[w];
        const bool is_cand = valid && many == 0;
        // apex of the triangle that kills an apparent edge at once (phase c): ANY common neighbour is valid (the other
        // triangles are verified in phase d); prefer one whose two edges predate the chunk, so that psi[a,v*] and
        // psi[b,v*] are final already and no in-chunk dependency arises.  Chosen here, with the ranks of its two edges,
        // while the masks are at hand: the two look-ups overlap the other waves' turns, and M0 dies here
        int vstar = 0, d1 = 0, d2 = 0, q1 = 0, q2 = 0;
        {
            u64 many0 = 0;
#pragma unroll
            for (int w = 0; w < NVW; ++w) many0 |= M0[w];
            if (many0) {
                if (NVW == 1 || M0[0]) vstar = __builtin_ctzll(M0[0]);
                else vstar = 64 + __builtin_ctzll(M0[NVW - 1]);
            } else if (many) {
                if (NVW == 1 || M[0]) vstar = __builtin_ctzll(M[0]);
                else vstar = 64 + __builtin_ctzll(M[NVW - 1]);
            }
            if (valid && many != 0) {
                d1 = pair_index(a, vstar); d2 = pair_index(b, vstar);
                q1 = (int)rank[d1] - r0; q2 = (int)rank[d2] - r0;   // < tid
                if (TDA_FAULT(2) && tid == 77 && q1 >= 0) q1 = tid;
            }
        }
        PROF_MARK(4);
        PROF_STOPC(11, out_k0 = k0; out_k1 = k1; out_status = 0);
        // ---- b. candidates (edges without a common neighbour): Kruskal in rank order ----
        // Only the union-find itself is sequential: wave 0 walks the candidates with the component labels
        // in registers and leaves one "merge" bit per edge.  Everything that follows from the decision
        // -- H0 rows, class bits of the births, psi/brank/bkey entries, counters -- is allocated by
        // prefix sums over ballots and written by the lanes that own the edges.
        u64* mbal = reinterpret_cast<u64*>(misc + MISC_COMP);        // [8] merge bits per wave of edges
        u64* zbal = mbal + 8;                                        // [8] candidates of length zero (no H0 row)
        int* offs = reinterpret_cast<int*>(mbal + 16);               // [8] births, [8] rows before each wave, totals, clen
        u16* freebits = reinterpret_cast<u16*>(misc + MISC_CKEY);    // j-th free class index
        float mykey = 0.f;
        if (is_cand) mykey = keyfn(r, a, b);         // every candidate needs its length (H0 death / H1 birth)
        {
            const u64 bal = __ballot(is_cand), zb = __ballot(is_cand && mykey == 0.0f);
            if (lane == 0) { cand[wave] = bal; zbal[wave] = zb; }
        }
        done[tid] = 0;
        __syncthreads();
        PROF_MARK(21);
        PROF_STOPC(12, out_k0 = k0; out_k1 = k1; out_status = 0);
        int nfree = 0;
#pragma unroll
        for (int c = 0; c < W; ++c) nfree += __builtin_popcountll((u64)(WT)~alive[c]);
        // Many candidates (the first chunks, where the forest grows): Boruvka rounds by the whole workgroup instead
        // of the sequential walk.  Every component picks its earliest incident candidate (cut property: with
        // distinct ranks it is a forest edge, i.e. a merge of Kruskal's order); the picked edges are contracted by
        // pointer jumping; candidates whose ends meet in one component without having been picked are births.
        // O(log n) rounds whatever the order of the edges.  The labels travel through LDS for the occasion.
        int ncand = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) ncand += __builtin_popcountll(cand[w]);
        int btot = 0;                                            // births of this chunk
        if (ncand > 0) {                                         // (a chunk without candidates has no phase b at all)
        const bool par = ncand > 24;                             // (48: the same, 96: 1 % slower)
        const int lblA0 = compA, lblB0 = compB;                  // labels at chunk start (wave 0)
        int qpar = NT;                                           // cut of the chunk found by the parallel path
        if (par) {
            int* bcomp = reinterpret_cast<int*>(misc + MISC_LIST + 1024);    // [128] label of every vertex
            u32* bbest2 = reinterpret_cast<u32*>(bcomp + 128);               // [2][128] earliest candidate at a label,
            int* bpar = reinterpret_cast<int*>(bcomp + 384);                 //          one set per round parity
                                                                             // [128] contraction pointers
            // classify the candidates at chunk positions < limit, starting from the labels of the chunk start
            auto rounds = [&](int limit) {
                if (wave == 0) { bcomp[lane] = lblA0; bcomp[64 + lane] = lblB0; }
                if (tid < 256) bbest2[tid] = 0xffffffffu;
                __syncthreads();
                bool live = is_cand && tid < limit, picked = false;
                for (int round = 0; round < 64; ++round) {
                    u32* bbest = bbest2 + 128 * (round & 1);
                    int ca = 0, cb2 = 0;
                    if (live) {
                        ca = bcomp[a]; cb2 = bcomp[b];
                        if (ca == cb2) live = false;                          // ends already joined: a birth
                        else { atomicMin(&bbest[ca], (u32)tid); atomicMin(&bbest[cb2], (u32)tid); }
                    }
                    if (!wg_any<NT>(vote, live)) break;
                    if (live && (bbest[ca] == (u32)tid || bbest[cb2] == (u32)tid)) { picked = true; live = false; }
                    // The contraction is the business of ONE wave, two labels per lane, and needs no barrier inside:
                    // the LDS serves the accesses of a wave in program order.  Nobody else touches bcomp / bpar between
                    // the vote above and the barrier below; this round's bbest is still being read by the others, so
                    // it is left alone and the OTHER set (last read a round ago) is wiped for the next round.
                    if (wave == 0) {
                        TDA_SOLO_BEGIN();
                        int pp[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int me = lane + 64 * h;
                            pp[h] = me;
                            const u32 bp = bbest[me];
                            if (bp != 0xffffffffu) {                          // point at the label on the other side of my edge
                                const u32 pk = ord[r0 + (int)bp];
                                const int oa = bcomp[pk >> 8], ob = bcomp[pk & 255u];
                                pp[h] = oa == me ? ob : oa;
                                // two labels that picked the same edge would point at each other: the smaller one is the root
                                if (bbest[pp[h]] == bp && me < pp[h]) pp[h] = me;
                            }
                        }
                        bpar[lane] = pp[0]; bpar[64 + lane] = pp[1];
                        for (int it = 0; it < 8; ++it) {                      // pointer jumping
                            const int p0 = bpar[lane], p1 = bpar[64 + lane];
                            const int g0 = bpar[p0], g1 = bpar[p1];
                            if (__ballot(g0 != p0 || g1 != p1) == 0ull) break;
                            bpar[lane] = g0; bpar[64 + lane] = g1;
                        }
                        const int c0 = bcomp[lane], c1 = bcomp[64 + lane];
                        const int n0 = bpar[c0], n1 = bpar[c1];
                        bcomp[lane] = n0; bcomp[64 + lane] = n1;
                        u32* other = bbest2 + 128 * ((round + 1) & 1);
                        other[lane] = 0xffffffffu; other[64 + lane] = 0xffffffffu;
                        TDA_SOLO_END();
                    }
                    __syncthreads();
                }
                const u64 mb = __ballot(picked);
                if (lane == 0) mbal[wave] = mb;
                __syncthreads();
            };
            rounds(NT);
            // do the births fit the free class bits?  If not, the chunk ends just before the first birth that does
            // not fit (Kruskal's decisions for the edges before it do not depend on the later ones): classify again
            // up to there, so that the labels contain exactly the merges of the shortened chunk
            int nbirth = 0, wcut = -1, before = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) {
                const int nb = __builtin_popcountll(cand[w] & ~mbal[w]);
                if (wcut < 0 && nbirth + nb > nfree) { wcut = w; before = nbirth; }
                nbirth += nb;
            }
            if (wcut >= 0) {
                u64 bw = cand[wcut] & ~mbal[wcut];
                for (int k = before; k < nfree; ++k) bw &= bw - 1ull;          // drop the births that still fit
                qpar = 64 * wcut + __builtin_ctzll(bw);
                PROF_COUNT(26, 1);
                __syncthreads();                                               // every thread has read the ballots
                rounds(qpar);
            }
            if (wave == 0) { compA = bcomp[lane]; compB = bcomp[64 + lane]; }
        }
        if (wave == 0) {
            TDA_SOLO_BEGIN();
            // The walk stops at the first birth that finds no free class bit: the chunk is closed just before
            // that edge, so that the kills of the shortened chunk can free bits (capacity = classes alive at
            // once).  q = NT: the whole chunk went through.
            const u64 candv = cand[lane & 7];                 // all ballots in one LDS read
            int q = par ? qpar : NT, births = 0;
            const bool walk = !par;
            for (int g = 0; g < NT / 64 && walk; ++g) {
                u64 cb = q == NT ? rl64(candv, g) : 0ull;
                u64 mm = 0ull;
                if (cb) {
                    PROF_COUNT(20, __builtin_popcountll(cb));
                    const u32 pkv = ord[r0 + 64 * g + lane < E ? r0 + 64 * g + lane : E - 1];
                    while (cb) {
                        const int l = __builtin_ctzll(cb);
                        cb &= cb - 1;
                        const u32 pk = rl32(pkv, l);
                        const int qa = (int)(pk >> 8), qb = (int)(pk & 255u);
                        int ca, cbb;
                        if (NVW == 1) {
                            ca = (int)rl32((u32)compA, qa);
                            cbb = (int)rl32((u32)compA, qb);
                        } else {
                            const int a0 = (int)rl32((u32)compA, qa & 63), a1 = (int)rl32((u32)compB, qa & 63);
                            const int b0 = (int)rl32((u32)compA, qb & 63), b1 = (int)rl32((u32)compB, qb & 63);
                            ca = qa < 64 ? a0 : a1;
                            cbb = qb < 64 ? b0 : b1;
                        }
                        // branch-free: once q is set the remaining edges of the group change nothing
                        const bool same = ca == cbb, open = q == NT;
                        q = (open && same && births == nfree) ? 64 * g + l : q;
                        births += same ? 1 : 0;
                        // relabel the component of b (a no-op when both ends carry the same label already)
                        const int cnew = (q == NT) ? ca : cbb;
                        compA = compA == cbb ? cnew : compA;
                        if (NVW == 2) compB = compB == cbb ? cnew : compB;
                        mm |= (!same && q == NT) ? (1ull << l) : 0ull;
                    }
                }
                if (lane == 0) mbal[g] = mm;
            }
            // births / rows before each wave of edges and the totals (lane g looks at group g)
            int nb = 0, nr = 0, nm = 0;
            if (lane < NT / 64) {
                u64 cg = cand[lane];
                if (64 * lane >= q) cg = 0ull;
                else if (q < 64 * lane + 64) cg &= (1ull << (q - 64 * lane)) - 1ull;
                const u64 mg = mbal[lane], zg = zbal[lane];
                nb = __builtin_popcountll(cg & ~mg); nr = __builtin_popcountll(mg & ~zg); nm = __builtin_popcountll(mg);
            }
            int pb = 0, pr = 0, tb = 0, tr = 0, tm = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) {
                const int xb = (int)rl32((u32)nb, w), xr = (int)rl32((u32)nr, w), xm = (int)rl32((u32)nm, w);
                if (w < lane) { pb += xb; pr += xr; }
                tb += xb; tr += xr; tm += xm;
            }
            if (lane < NT / 64) { offs[lane] = pb; offs[8 + lane] = pr; }
            if (lane == 0) { offs[16] = tb; offs[17] = tr; offs[18] = tm; offs[19] = q; }
            TDA_SOLO_END();
        } else {
            // meanwhile: the j-th free class index (every class looks up its own rank among the free ones)
            for (int t = tid - 64; t < WB * W; t += NT - 64) {
                const int cw = t / WB, bit = t & (WB - 1);
                int rk = 0;
                bool fr = false;
#pragma unroll
                for (int c = 0; c < W; ++c) {
                    const WT f_ = (WT)~alive[c];
                    if (c < cw) rk += __builtin_popcountll((u64)f_);
                    if (c == cw) { rk += __builtin_popcountll((u64)(f_ & (((WT)1 << bit) - (WT)1))); fr = (f_ >> bit) & (WT)1; }
                }
                if (fr) freebits[rk] = (u16)t;
            }
        }
        __syncthreads();
        PROF_MARK(22);
        PROF_STOPC(13, out_k0 = k0; out_k1 = k1; out_status = 0);
        clen = uni(offs[19]);                                 // (workgroup-uniform values read from LDS: to SGPRs)
        if (clen == 0) status |= TDA_WIN_CLASS_OVERFLOW;      // not even the first edge of the chunk fits
        if (clen < NT) PROF_COUNT(14, 1);
        bool mrg = is_cand && ((mbal[wave] >> lane) & 1ull);
        bool birth = is_cand && !mrg && tid < clen;
        bool row = mrg && mykey != 0.0f;                           // zero-length edges give no row
        const u64 below = (1ull << lane) - 1ull;
        const int bpre = offs[wave] + __builtin_popcountll(__ballot(birth) & below);
        const int rpre = offs[8 + wave] + __builtin_popcountll(__ballot(row) & below);
        btot = uni(offs[16]);
        const int rtot = uni(offs[17]), mtot = uni(offs[18]);
        if (!status) {
            if (row) {
                const int pos = k0 + rpre;
                if (pos < h0_cap) { h0[2 * pos] = 0.0; h0[2 * pos + 1] = (double)mykey; }
            }
            k0 += rtot; merges += mtot;
            if (birth) {
                const int idx = (int)freebits[bpre];
                const int cw = idx / WB, bit = idx & (WB - 1);
                Psi<W, WT> nv = pzero<W, WT>();
#pragma unroll
                for (int c = 0; c < W; ++c)
                    if (c == cw) nv.w[c] = ((WT)1 << bit);
                if (!NARROW || r < psi_cap) psi[slot] = nv;
                brank[idx] = r;
                bkey[idx] = mykey;
            }
            // one class word: wave 0 keeps the classes alive in birth order, one per lane, so that the elder rule of the
            // kill reduction is a ballot and a count of leading zeros.  The births of a chunk come in rank order and
            // are younger than everything alive: they are appended
            if (W == 1 && wave == 0 && btot > 0) {
                const int nal = __builtin_popcountll((u64)alive[0]);
                if (lane >= nal && lane < nal + btot) ordcls = (int)freebits[lane - nal];
            }
            // the btot lowest free class indices are in use now (every thread updates its copy)
            int left = btot;
#pragma unroll
            for (int c = 0; c < W; ++c) {
                const u64 f_ = (u64)(WT)~alive[c];
                const int pc = __builtin_popcountll(f_);
                if (left >= pc) { alive[c] = (WT)~(WT)0; left -= pc; }
                else if (left > 0) {
                    int lo = 0, hi = 63;                   // smallest p with `left` free bits at positions <= p
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (__builtin_popcountll(f_ & ((2ull << mid) - 1ull)) >= left) hi = mid; else lo = mid + 1;
                    }
                    alive[c] |= (WT)(f_ & ((2ull << lo) - 1ull));
                    left = 0;
                }
            }
        }
        }   // ncand > 0
        if (status) break;
        PROF_MARK(5);
        PROF_STOPC(14, out_k0 = k0; out_k1 = k1; out_status = 0);
        // No class alive and none born in this chunk: every psi entry is zero (dead classes were substituted out,
        // unwritten entries start at zero), so the apparent edges of the chunk get the zero vector they already hold
        // and no triangle can kill anything.  Phases c and d are skipped; only the adjacency rows move on.  This is
        // the state of the long-edge tail of a filtration.
        bool quiet = btot == 0;
#pragma unroll
        for (int c = 0; c < W; ++c) quiet = quiet && alive[c] == 0;
        u32* lcnt = reinterpret_cast<u32*>(misc + MISC_MIN);         // list header: [0] entries, [1] earliest key
        if (quiet) PROF_COUNT(24, 1);
#ifdef TDA_PROFILE
        if (!quiet) prof_rneed = r0 + clen;          // class vectors are needed for the ranks below this
#endif
        if constexpr (NARROW) {
            // a busy chunk whose class vectors would reach into ord[r0 ...]: for the wide pass
            if (!quiet && r0 + NT > psi_cap) { status |= TDA_WIN_CLASS_OVERFLOW; break; }
        }
        if (!quiet) {
        // ---- c. apparent edges: psi[e] = psi[a,v*] ^ psi[b,v*] ----
        // (the apex v* and the ranks of its two edges were found in phase a)
        const bool apparent = valid && many != 0 && tid < clen;
        bool pending = apparent;
        Psi<W, WT> base = pzero<W, WT>();
        // (NARROW: the vector of an edge is found at its rank, which is what q1 / q2 hold)
        const int s1 = NARROW ? q1 + r0 : d1, s2 = NARROW ? q2 + r0 : d2;
        if (apparent && q1 < 0 && q2 < 0) {  // both edges predate the chunk (see v*): final already, no waiting
            base = pxor(psi[s1], psi[s2]);
            psi[slot] = base;
            pending = false;
        }
        // The rest waits for its two edges, WITHOUT barriers: dependencies point at earlier edges of the chunk only,
        // every wave of the workgroup is resident, and the LDS serves the accesses of a wave in program order -- so an
        // owner writes its vector, then its flag, and a reader that has seen the flag reads the vector.  A wave polls
        // until its own lanes are settled (a chain inside the wave advances one link per trip); the barrier after the
        // loop is the only one of the phase (it used to be two per dependency level: 12 to 29 levels per window).
        volatile unsigned char* vdone = done;
        LDS_ORDER();       // (births and the lanes just above wrote their vectors)
        if (!pending) vdone[tid] = 1;        // candidates, idle lanes and the lanes above are settled
#ifdef TDA_PROFILE
#ifndef TDA_PROFILE_STOPS_ONLY
        if (pending) atomicAdd(&prof_lds[33], 1ull);
#endif
        PROF_COUNT(34, 1);
#endif
        int idle_trips = 0;
        while (__ballot(pending)) {
            PROF_COUNT(32, 1);
            bool ready = false;
            if (pending) ready = (q1 < 0 || vdone[q1]) && (q2 < 0 || vdone[q2]);
            const u64 rbal = __ballot(ready);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (ready) {
                base = pxor(psi[s1], psi[s2]);
                psi[slot] = base;
                LDS_ORDER();
                vdone[tid] = 1;
                pending = false;
            }
            if (!rbal) {
                if (++idle_trips > TDA_POLL_LIMIT) {                 // give up: poison, and let the waiters on MY lanes through
                    if (lane == 0) reinterpret_cast<volatile u32*>(misc + MISC_MIN)[3] = 1u;
                    if (pending) vdone[tid] = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);                         // nothing to do this trip: leave the issue slots to the others (0 / 1 / 2 / 4: no difference measured)
            }
        }
        __syncthreads();
        PROF_MARK(6);
        PROF_STOPC(15, out_k0 = k0; out_k1 = k1; out_status = 0);
        // ---- d. the other triangles of every apparent edge ----
        // Link argument: if v and v' are common neighbours of (a,b) and adjacent to each other, the
        // tetrahedron (a,b,v,v') shows that triangles (a,b,v) and (a,b,v') carry the same boundary
        // class (its other two faces entered earlier, so their boundaries are already trivial).
        // Hence only common neighbours OUTSIDE the connected component of v* in the link need a
        // test.  The closure uses the adjacency rows of the chunk start (a subset of the true
        // adjacency: it can only leave more vertices to test, never fewer than necessary).
        u32 m0, m1, m2 = 0u, m3 = 0u;
        {
            u64 rem[NVW];
#pragma unroll
            for (int w = 0; w < NVW; ++w) rem[w] = 0ull;
            if (apparent) {
                // component of v* in the link: one push from v*, then PULL -- a remaining vertex joins as
                // soon as one of its neighbours is in the component.  The remaining set is small (mostly
                // empty after the push), its adjacency reads are independent of each other, and four are
                // in flight per trip; a push over the (large) frontier would be one dependent LDS read
                // per vertex.
                u64 comp[NVW];
#pragma unroll
                for (int w = 0; w < NVW; ++w) comp[w] = 0ull;
                if (NVW == 1 || vstar < 64) comp[0] = 1ull << vstar;
                else comp[NVW - 1] = 1ull << (vstar - 64);
#pragma unroll
                for (int w = 0; w < NVW; ++w) { comp[w] |= adj[2 * vstar + w] & M[w]; rem[w] = M[w] & ~comp[w]; }
                bool changed = true;
                while (changed) {
                    changed = false;
#pragma unroll
                    for (int w = 0; w < NVW; ++w) {
                        u64 rr = rem[w];
                        while (rr) {
                            int u[4]; bool ok[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                ok[k] = rr != 0ull;
                                u[k] = ok[k] ? 64 * w + __builtin_ctzll(rr) : vstar;
                                rr &= rr - 1ull;
                            }
                            u64 nb[4][NVW];
#pragma unroll
                            for (int k = 0; k < 4; ++k)
#pragma unroll
                                for (int x = 0; x < NVW; ++x) nb[k][x] = adj[2 * u[k] + x];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                u64 hit = 0ull;
#pragma unroll
                                for (int x = 0; x < NVW; ++x) hit |= nb[k][x] & comp[x];
                                if (ok[k] && hit) {
                                    comp[w] |= 1ull << (u[k] & 63);
                                    rem[w] &= ~(1ull << (u[k] & 63));
                                    changed = true;
                                }
                            }
                        }
                    }
                    u64 anyr = 0ull;
#pragma unroll
                    for (int w = 0; w < NVW; ++w) anyr |= rem[w];
                    if (!anyr) break;
                }
            }
            m0 = (u32)rem[0]; m1 = (u32)(rem[0] >> 32);
            if (NVW == 2) { m2 = (u32)rem[NVW - 1]; m3 = (u32)(rem[NVW - 1] >> 32); }
        }
        PROF_MARK(16);
        PROF_STOPC(16, out_k0 = k0; out_k1 = k1; out_status = 0);
        // All non-trivial triangles of the chunk are listed under the frozen table (key = lane << 8 | v, the
        // order in which a sequential sweep meets them) and wave 0 reduces the list in registers: the
        // earliest non-zero vector kills the YOUNGEST class in it (elder rule) and is substituted into
        // the remaining vectors; the substitutions are linear, so triangles that were trivial stay
        // trivial and the list is complete.  The psi table is rewritten ONCE per chunk with the
        // composed substitution.  If the list does not fit, only its earliest entry is processed and
        // the chunk is listed again.
        const int ta_ = tri2(a), tb_ = tri2(b);
        constexpr int ES = 4 + W * (int)sizeof(WT);                  // list entry: key, vector
        // One class word: wave 0 puts the list into the order of the killing edges first (counting sort on the lane
        // that listed the entry: every thread leaves its number of entries in hist[], entry keys carry their index
        // within the thread in bits 24..31), so that "the earliest non-zero vector" is a ballot and a count of
        // trailing zeros instead of a reduction over the wave.  Entries of ONE edge may come in any order: the
        // classes they kill and the images of those classes depend on their span only.
        constexpr bool SORTED = W == 1;
        constexpr int HISTN = (NT + 255) & ~255;                     // (counters beyond NT stay zero: the scan reads whole words)
        constexpr int HIST_BYTES = SORTED ? HISTN + 8 : 0;           // u8 hist[NT] behind the entries (8-byte aligned)
        constexpr int LMAX = SORTED ? 255 : 256;                     // (sorted: offsets are bytes)
        constexpr int LCAP = ((MISC_LIST_BYTES - HIST_BYTES) / ES) < LMAX ? ((MISC_LIST_BYTES - HIST_BYTES) / ES) : LMAX;
        constexpr int LPL = (LCAP + 63) / 64;
        // class-indexed image table (one Psi per class) when it fits the list area; else kill records
        constexpr bool FTAB = WB * W * (int)sizeof(Psi<W, WT>) <= MISC_LIST_BYTES;
        constexpr int KMAX = FTAB ? 64 : LCAP;                       // kills per reduction round
        unsigned char* list = misc + MISC_LIST;
        unsigned char* hist = list + ((LCAP * ES + 7) & ~7);
        const u32 mws[4] = {m0, m1, m2, m3};
#ifdef TDA_PROFILE
        {   // diagnostic: triangles left to test after the link closure, and apparent edges
            const int ntri = __builtin_popcount(m0) + __builtin_popcount(m1) + __builtin_popcount(m2) + __builtin_popcount(m3);
#ifndef TDA_PROFILE_STOPS_ONLY
            if (ntri) atomicAdd(&prof_lds[27], (unsigned long long)ntri);
            if (apparent) atomicAdd(&prof_lds[28], 1ull);
            if (ntri) atomicAdd(&prof_lds[29], 1ull);
#endif
        }
#endif
        int list_rounds = 0;
        while (true) {
            if (list_rounds > 0) {                     // the first listing finds the header reset (chunk end)
                if (tid == 0) { lcnt[0] = 0u; lcnt[1] = 0xffffffffu; }
                __syncthreads();
            }
            u32 firstkey = 0xffffffffu;
            Psi<W, WT> firsty = pzero<W, WT>();
            u32 mycnt = 0u;                                          // entries of this thread in this round
            if (apparent) {
                base = psi[slot];
                Psi<W, WT> prev = pzero<W, WT>();                    // a repeated vector reduces to zero: skip it
                // (64 vertices per word: the four triangles of a trip may come from anywhere in it)
#pragma unroll
                for (int wi = 0; wi < NVW; ++wi) {
                    u64 mm = (u64)mws[2 * wi] | ((u64)mws[2 * wi + 1] << 32);
                    const int vbase = 64 * wi;
                    while (mm) {
                        const int v0 = vbase + __builtin_ctzll(mm); mm &= mm - 1ull;
                        const bool ok1 = mm != 0ull; const int v1 = ok1 ? vbase + __builtin_ctzll(mm) : v0; mm &= mm - 1ull;
                        const bool ok2 = mm != 0ull; const int v2 = ok2 ? vbase + __builtin_ctzll(mm) : v0; mm &= mm - 1ull;
                        const bool ok3 = mm != 0ull; const int v3 = ok3 ? vbase + __builtin_ctzll(mm) : v0; mm &= mm - 1ull;
#define TDA_IDX_(v, ja, jb)                                                       \
                        const int t##ja = (int)(__umul24((u32)(v), (u32)((v) - 1)) >> 1); \
                        const int f##ja = (v) < a ? ta_ + (v) : t##ja + a;                 \
                        const int f##jb = (v) < b ? tb_ + (v) : t##ja + b;                 \
                        const int ja = NARROW ? (int)rank[f##ja] : f##ja;                  \
                        const int jb = NARROW ? (int)rank[f##jb] : f##jb;
                        TDA_IDX_(v0, ja0, jb0) TDA_IDX_(v1, ja1, jb1) TDA_IDX_(v2, ja2, jb2) TDA_IDX_(v3, ja3, jb3)
#undef TDA_IDX_
                        const Psi<W, WT> p0 = psi[ja0], q0 = psi[jb0], p1 = psi[ja1], q1_ = psi[jb1];
                        const Psi<W, WT> p2 = psi[ja2], q2_ = psi[jb2], p3 = psi[ja3], q3 = psi[jb3];
                        const Psi<W, WT> ys[4] = {pxor(pxor(p0, q0), base), pxor(pxor(p1, q1_), base),
                                                  pxor(pxor(p2, q2_), base), pxor(pxor(p3, q3), base)};
                        const bool oks[4] = {true, ok1, ok2, ok3};
                        const int vs[4] = {v0, v1, v2, v3};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (!oks[k] || !pnz(ys[k]) || !pnz(pxor(ys[k], prev))) continue;
                            prev = ys[k];
                            const u32 key = ((u32)tid << 8) | (u32)vs[k];
                            if (firstkey == 0xffffffffu) { firstkey = key; firsty = ys[k]; }
                            const u32 idx = atomicAdd(&lcnt[0], 1u);
                            if (idx < (u32)LCAP) {
                                *reinterpret_cast<u32*>(list + ES * idx) = SORTED ? key | (mycnt << 24) : key;
#pragma unroll
                                for (int c = 0; c < W; ++c)
                                    *reinterpret_cast<WT*>(list + ES * idx + 4 + c * (int)sizeof(WT)) = ys[k].w[c];
                            }
                            ++mycnt;
                        }
                    }
                }
            }
            if (SORTED) {
                hist[tid] = (unsigned char)mycnt;
                if (NT < HISTN && tid < HISTN - NT) hist[NT + tid] = 0;
            }
            {
                const u32 wmin = wave_min_u32_dpp(firstkey);
                if (lane == 0 && wmin != 0xffffffffu) atomicMin(&lcnt[1], wmin);
            }
            __syncthreads();
            PROF_MARK(17);
            PROF_STOPC(17, out_k0 = k0; out_k1 = k1; out_status = 0);
            const u32 cnt = (u32)uni((int)lcnt[0]);
            if (cnt == 0u) break;
            const bool complete = cnt <= (u32)LCAP;
            PROF_COUNT(11, 1);
            PROF_COUNT(12, cnt);
            if (!complete) {
                PROF_COUNT(13, 1);
                if (firstkey == lcnt[1]) {                            // exactly one lane: keys are unique
                    *reinterpret_cast<u32*>(list) = firstkey;
#pragma unroll
                    for (int c = 0; c < W; ++c) *reinterpret_cast<WT*>(list + 4 + c * (int)sizeof(WT)) = firsty.w[c];
                }
                __syncthreads();
            }
            WT alive_before[W];                                      // wave 0 updates `alive` while it reduces
#pragma unroll
            for (int c = 0; c < W; ++c) alive_before[c] = alive[c];
            if (wave == 0) {
                TDA_SOLO_BEGIN();
                const int nl = complete ? (int)cnt : 1;
                u32 ek[LPL];
                Psi<W, WT> ev[LPL];
#pragma unroll
                for (int q = 0; q < LPL; ++q) {
                    const int idx = lane + 64 * q;
                    ek[q] = 0xffffffffu; ev[q] = pzero<W, WT>();
                    if (idx < nl) {
                        ek[q] = *reinterpret_cast<const u32*>(list + ES * idx);
#pragma unroll
                        for (int c = 0; c < W; ++c)
                            ev[q].w[c] = *reinterpret_cast<const WT*>(list + ES * idx + 4 + c * (int)sizeof(WT));
                    }
                }
                if (SORTED && complete) {
                    // exclusive prefix over hist[] (HW packed words of four counts per lane; at most 255 entries, so
                    // bytes never carry), entries written back at offset-of-thread + index-in-thread, reloaded
                    constexpr int HW = HISTN / 256;
                    u32* hw = reinterpret_cast<u32*>(hist) + HW * lane;
                    const u32 x0 = hw[0], x1 = HW == 2 ? hw[HW - 1] : 0u;
                    const u32 s0 = __builtin_amdgcn_sad_u8(x0, 0u, 0u);
                    const u32 tot = HW == 2 ? __builtin_amdgcn_sad_u8(x1, 0u, s0) : s0;
                    const u32 before = (u32)wave_incl_scan_i32((int)tot) - tot;
                    hw[0] = x0 * 0x01010101u - x0 + before * 0x01010101u;
                    if (HW == 2) hw[HW - 1] = x1 * 0x01010101u - x1 + (before + s0) * 0x01010101u;
#pragma unroll
                    for (int q = 0; q < LPL; ++q) {
                        if (lane + 64 * q < nl) {
                            const int pos = (int)hist[(ek[q] >> 8) & 0xffffu] + (int)(ek[q] >> 24);
                            *reinterpret_cast<u32*>(list + ES * pos) = ek[q] & 0x00ffffffu;
                            *reinterpret_cast<WT*>(list + ES * pos + 4) = ev[q].w[0];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < LPL; ++q) {
                        const int idx = lane + 64 * q;
                        if (idx < nl) {
                            ek[q] = *reinterpret_cast<const u32*>(list + ES * idx);
                            ev[q].w[0] = *reinterpret_cast<const WT*>(list + ES * idx + 4);
                        }
                    }
                }
                // birth rank / length of the classes: lane i keeps bit i of every word
                const int lb = lane & (WB - 1);
                int brk[W];
                float bky[W];
#pragma unroll
                for (int c = 0; c < W; ++c) {
                    brk[c] = lane < WB ? brank[WB * c + lb] : -1;
                    bky[c] = lane < WB ? bkey[WB * c + lb] : 0.f;
                }
                // FTAB: lane j keeps kill j -- the image of its class under all LATER kills (composed
                // substitution), the rank of the killing edge and the birth length
                int nalive0 = W == 1 ? __builtin_popcountll((u64)alive[0]) : 0;     // lanes of `ordcls` in use
                Psi<W, WT> vimg = pzero<W, WT>();
                u32 mycode = 0u, myrk = 0u;
                float mybirth = 0.f;
                int nk = 0, more = 0;
                const int nq = (nl + 63) >> 6;                        // register blocks the list reaches (wave-uniform)
                PROF_MARK(30);
                if constexpr (SORTED) {
                    // Block by block in list order.  Lane j keeps kill j (class bit, vector at the time of the kill);
                    // a block first catches up with the kills of the blocks before it, then its earliest non-zero
                    // vector kills, is substituted into the rest of ITS block, and so on: one ballot, one count of
                    // trailing zeros and two lane reads per kill on the critical path.
                    WT kvec = 0;
                    // lane i: mask of the i-th oldest class alive (0 beyond the classes in use).  A class that is
                    // killed leaves a zero in its lane: the order of the others stands, and the lanes are closed up
                    // ONCE after the round (a shift of the order per kill was three DPP moves and four selects on a
                    // wave that issues one vector instruction every eight cycles, profiles/r03_valu_rate.txt)
                    WT cm = lane < nalive0 ? (WT)1 << ordcls : (WT)0;
                    bool stop = false;
#pragma unroll
                    for (int q = 0; q < LPL; ++q) {
                        if (q >= nq || stop) break;
                        WT v = ev[q].w[0];
                        const u32 myk = ek[q];
                        for (int k = 0; k < nk; ++k) {                 // the kills so far, in their order
                            const WT m = (WT)1 << rl32(mycode, k);
                            const WT w = sizeof(WT) == 8 ? (WT)rl64((u64)kvec, k) : (WT)rl32((u32)kvec, k);
                            if (v & m) v ^= w;
                        }
                        u64 nz = __ballot(v != 0);
                        while (nz) {
                            if (nk == KMAX) { more = 1; stop = true; break; }
                            const int src = __builtin_ctzll(nz);
                            const WT wvw = sizeof(WT) == 8 ? (WT)rl64((u64)v, src) : (WT)rl32((u32)v, src);
                            // elder rule: the youngest member is the highest lane whose class is in the vector
                            const u64 ob = __ballot((cm & wvw) != 0);
                            if (!ob) { status |= TDA_WIN_CLASS_OVERFLOW; stop = true; break; }       // cannot happen: never spin
                            const int opos = 63 - __builtin_clzll(ob);
                            const int ybit = (int)rl32((u32)ordcls, opos);
                            const WT ybm = (WT)1 << ybit;
                            if (v & ybm) v ^= wvw;                      // (the killer itself becomes zero)
                            nz = __ballot(v != 0);
                            // ---- off the critical path ----
                            const u32 kk = rl32(myk, src);
                            const float ybirth = __uint_as_float(rl32(__float_as_uint(bky[0]), ybit));
                            cm &= (WT)~ybm;                             // the class leaves the order (its lane: zero)
                            // lane j keeps kill j: the image of its class under all LATER kills (lanes beyond hold zero)
                            const bool me = lane == nk;
                            WT vi = vimg.w[0];
                            if (vi & ybm) vi ^= wvw;
                            vimg.w[0] = me ? (WT)(wvw & (WT)~ybm) : vi;
                            mycode = me ? (u32)ybit : mycode;
                            myrk = me ? (u32)(r0 + (int)((kk >> 8) & 0xffffu)) : myrk;
                            mybirth = me ? ybirth : mybirth;
                            kvec = me ? wvw : kvec;
                            alive[0] &= (WT)~ybm;
                            ++nk;
                        }
                    }
                    if (nk > 0) {
                        // close up the order: the classes still alive move to lanes 0 .. in their order, the other lanes
                        // go behind them (a permutation: every lane is written)
                        const bool keep = cm != 0;
                        const u64 keepb = __ballot(keep);
                        const int below = (int)__builtin_amdgcn_mbcnt_hi((u32)(keepb >> 32), __builtin_amdgcn_mbcnt_lo((u32)keepb, 0u));
                        const int dst = keep ? below : __builtin_popcountll(keepb) + (lane - below);
                        ordcls = __builtin_amdgcn_ds_permute(dst << 2, ordcls);
                    }
                } else
                while (true) {
                    u32 mk = 0xffffffffu;
#pragma unroll
                    for (int q = 0; q < LPL; ++q)
                        if (q < nq && pnz(ev[q]) && ek[q] < mk) mk = ek[q];
                    const u32 best = wave_min_u32_dpp(mk);
                    if (best == 0xffffffffu) break;
                    if (nk == KMAX) { more = 1; break; }
                    const int src = __builtin_ctzll(__ballot(mk == best));
                    Psi<W, WT> mine = pzero<W, WT>();
#pragma unroll
                    for (int q = 0; q < LPL; ++q)
                        if (q < nq && ek[q] == best) mine = ev[q];
                    Psi<W, WT> wv;
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        if (sizeof(WT) == 8) wv.w[c] = (WT)rl64((u64)mine.w[c], src);
                        else wv.w[c] = (WT)rl32((u32)mine.w[c], src);
                    }
                    const int rk = r0 + (int)(best >> 8);
                    // youngest class of wv (elder rule)
                    int ybit, ycw;
                    float ybirth;
                    if (W == 1) {
                        // lanes hold the classes alive in birth order: the youngest member is the highest lane set
                        const u64 ob = __ballot(lane < nalive0 && (((u64)wv.w[0] >> ordcls) & 1ull));
                        if (!ob) { status |= TDA_WIN_CLASS_OVERFLOW; break; }                    // cannot happen: never spin
                        const int opos = 63 - __builtin_clzll(ob);
                        ybit = (int)rl32((u32)ordcls, opos);
                        ycw = 0;
                        ybirth = __uint_as_float(rl32(__float_as_uint(bky[0]), ybit));
                        // the class leaves the order: the younger ones move down one lane
                        const int up = __builtin_amdgcn_update_dpp(0, ordcls, 0x130, 0xF, 0xF, true);     // wave_shl:1 = lane + 1
                        if (lane >= opos) ordcls = up;
                        --nalive0;
                    } else {
                        int candv = -1, cwl = 0;
                        float byl = 0.f;
#pragma unroll
                        for (int c = 0; c < W; ++c)
                            if (lane < WB && ((wv.w[c] >> lb) & (WT)1) && brk[c] > candv) { candv = brk[c]; cwl = c; byl = bky[c]; }
                        const int bestr = wave_max_i32_dpp(candv);
                        const u64 ybal = __ballot(candv == bestr && candv >= 0);                 // birth ranks are distinct
                        if (!ybal) { status |= TDA_WIN_CLASS_OVERFLOW; break; }                  // cannot happen: never spin
                        ybit = __builtin_ctzll(ybal);
                        ycw = (int)rl32((u32)cwl, ybit);
                        ybirth = __uint_as_float(rl32(__float_as_uint(byl), ybit));
                    }
                    if (FTAB) {
                        WT sel = 0;
#pragma unroll
                        for (int c = 0; c < W; ++c)
                            if (c == ycw) sel = (vimg.w[c] >> ybit) & (WT)1;
                        if (lane < nk && sel) vimg = pxor(vimg, wv);
                        if (lane == nk) {
                            vimg = wv;
#pragma unroll
                            for (int c = 0; c < W; ++c)
                                if (c == ycw) vimg.w[c] &= (WT)~((WT)1 << ybit);
                            mycode = ((u32)ycw << 8) | (u32)ybit; myrk = (u32)rk; mybirth = ybirth;
                        }
                    } else {
                        const u32 pkk = ord[rk];
                        const float key = keyfn(rk, (int)(pkk >> 8), (int)(pkk & 255u));
                        if (key > ybirth) {
                            if (k1 < h1_cap && lane == 0) { h1[2 * k1] = (double)ybirth; h1[2 * k1 + 1] = (double)key; }
                            ++k1;
                        }
                        if (lane == 0) {                              // kill record nk (the list is in registers by now)
                            *reinterpret_cast<u32*>(list + ES * nk) = ((u32)ycw << 8) | (u32)ybit;
#pragma unroll
                            for (int c = 0; c < W; ++c) *reinterpret_cast<WT*>(list + ES * nk + 4 + c * (int)sizeof(WT)) = wv.w[c];
                        }
                    }
                    // substitute into the vectors still waiting (the killer itself becomes zero)
#pragma unroll
                    for (int q = 0; q < LPL; ++q) {
                        if (q >= nq) break;
                        WT sel = 0;
#pragma unroll
                        for (int c = 0; c < W; ++c)
                            if (c == ycw) sel = (ev[q].w[c] >> ybit) & (WT)1;
                        if (sel) ev[q] = pxor(ev[q], wv);
                    }
#pragma unroll
                    for (int c = 0; c < W; ++c)
                        if (c == ycw) alive[c] &= (WT)~((WT)1 << ybit);
                    ++nk;
                }
                PROF_MARK(31);
                PROF_COUNT(10, nk);                                  // (counted here: a counter update per kill would be a third of the loop)
                if (FTAB) {
                    // diagram rows of this round, one kill per lane; the image table for the rewrite
                    const bool mine_ok = lane < nk;
                    float key = 0.f;
                    if (mine_ok) {
                        const u32 pkk = ord[(int)myrk];
                        key = keyfn((int)myrk, (int)(pkk >> 8), (int)(pkk & 255u));
                    }
                    const bool emit = mine_ok && key > mybirth;
                    const u64 bal = __ballot(emit);
                    const int pos = k1 + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
                    if (emit && pos < h1_cap) { h1[2 * pos] = (double)mybirth; h1[2 * pos + 1] = (double)key; }
                    k1 += __builtin_popcountll(bal);
                    if (mine_ok) {
                        Psi<W, WT>* ftab = reinterpret_cast<Psi<W, WT>*>(list);
                        ftab[WB * (int)(mycode >> 8) + (int)(mycode & 255u)] = vimg;
                    }
                }
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < W; ++c) shared->alive[c] = (u64)alive[c];
                    shared->k1 = k1; shared->nk = nk; shared->more = more; shared->status = status;
                }
                TDA_SOLO_END();
            }
            __syncthreads();
            PROF_MARK(18);
            PROF_STOPC(18, out_k0 = k0; out_k1 = k1; out_status = 0);
            WT kmask[W];
#pragma unroll
            for (int c = 0; c < W; ++c) {
                const WT na = sizeof(WT) == 8 ? (WT)uni64(shared->alive[c]) : (WT)uni((int)(u32)shared->alive[c]);
                kmask[c] = alive_before[c] & (WT)~na; alive[c] = na;
            }
            k1 = uni(shared->k1);
            const int nk = uni(shared->nk);
            const bool more = uni(shared->more) != 0;
            status = uni(shared->status);
            if (status) break;
            if (FTAB) {
                // ---- rewrite: p -> (p minus killed classes) ^ images of the killed classes it contained ----
                const Psi<W, WT>* ftab = reinterpret_cast<const Psi<W, WT>*>(list);
                // only edges that have entered the filtration can carry class bits: walk them by rank
                const int seen_end = r0 + clen < Ev ? r0 + clen : Ev;          // the last chunk may be short
#pragma unroll 4
                for (int rr = tid; rr < seen_end; rr += NT) {
                    const int e = NARROW ? rr : edge_flat(ord[rr]);
                    const Psi<W, WT> p = psi[e];
                    WT anyh = 0;
#pragma unroll
                    for (int c = 0; c < W; ++c) anyh |= p.w[c] & kmask[c];
                    if (!anyh) continue;
                    Psi<W, WT> q;
#pragma unroll
                    for (int c = 0; c < W; ++c) q.w[c] = p.w[c] & (WT)~kmask[c];
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        u64 hh = (u64)(p.w[c] & kmask[c]);
                        while (hh) {
                            const int bit = __builtin_ctzll(hh);
                            hh &= hh - 1ull;
                            q = pxor(q, ftab[WB * c + bit]);
                        }
                    }
                    psi[e] = q;
                }
            } else {
                // ---- the composed substitution, four kills per pass over the table ----
                for (int j0 = 0; j0 < nk; j0 += 4) {
                    u32 kc[4];
                    Psi<W, WT> kw[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int jj = j0 + j < nk ? j0 + j : j0;
                        kc[j] = j0 + j < nk ? *reinterpret_cast<const u32*>(list + ES * jj) : 0xffffffffu;
#pragma unroll
                        for (int c = 0; c < W; ++c) kw[j].w[c] = *reinterpret_cast<const WT*>(list + ES * jj + 4 + c * (int)sizeof(WT));
                    }
#pragma unroll 4
                    for (int rr = tid; rr < (r0 + clen < Ev ? r0 + clen : Ev); rr += NT) {
                        const int e = NARROW ? rr : edge_flat(ord[rr]);
                        Psi<W, WT> p = psi[e];
                        bool changed = false;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (kc[j] == 0xffffffffu) continue;
                            const int jcw = (int)(kc[j] >> 8), jbit = (int)(kc[j] & 255u);
                            WT sel = 0;
#pragma unroll
                            for (int c = 0; c < W; ++c)
                                if (c == jcw) sel = (p.w[c] >> jbit) & (WT)1;
                            if (sel) { p = pxor(p, kw[j]); changed = true; }
                        }
                        if (changed) psi[e] = p;
                    }
                }
            }
            PROF_MARK(19);
            PROF_STOPC(19, out_k0 = k0; out_k1 = k1; out_status = 0);
            if (complete && !more) break;
            if (++list_rounds > 4 * NT) { status |= TDA_WIN_CLASS_OVERFLOW; break; }   // every round kills >= 1 class: never reached
            __syncthreads();            // table rewritten before the chunk is listed again
        }
        }   // !quiet
        // the chunk's edges join the adjacency rows; the rows of the chunk's own edges (last read in phase a)
        // and the list header are reset for the next chunk under the same barrier
        if (valid && tid < clen) {
            atomicOr(reinterpret_cast<unsigned long long*>(&adj[2 * a + (b >> 6)]), 1ull << (b & 63));
            atomicOr(reinterpret_cast<unsigned long long*>(&adj[2 * b + (a >> 6)]), 1ull << (a & 63));
        }
        if (tid < 256) adjc[tid] = 0ull;
        if (tid == 0) {
            lcnt[0] = 0u; lcnt[1] = 0xffffffffu; lcnt[2] = 0u; shared->clen = 0x7fffffff;
            if (lcnt[3]) shared->status = TDA_WIN_NOT_CONVERGED;          // (sampled by ONE thread: the verdict is uniform)
        }
        __syncthreads();
        if (uni(shared->status) & TDA_WIN_NOT_CONVERGED) { status |= TDA_WIN_NOT_CONVERGED; break; }
        // The end of the story: no class is alive and EVERY remaining edge already has a common neighbour, now and
        // (adjacency only grows) at its own time.  Then no remaining edge is a candidate: no component can merge,
        // no class can be born, and with nothing alive nothing can die -- the rest of the filtration adds no row.
        if (quiet) {
            const int from = r0 + clen;
            bool covered = true;
            // every thread walks its own residue class of the remaining edges and remembers where it stopped: an edge
            // that is covered stays covered, so no edge is looked at twice over all the checks of a window
            int rr = (NT & (NT - 1)) == 0 ? from + ((tid - from) & (NT - 1)) : from + (((tid - from) % NT) + NT) % NT;
            rr = rr > cov_next ? rr : cov_next;
            while (rr < Ev) {
                const u32 pk = ord[rr];
                const int ea = (int)(pk >> 8), eb = (int)(pk & 255u);
                u64 common = adj[2 * ea] & adj[2 * eb];
                if (NVW == 2) common |= adj[2 * ea + 1] & adj[2 * eb + 1];
                if (common == 0ull) { covered = false; break; }
                rr += NT;
            }
            cov_next = rr;
            // the FIRST edge without a common neighbour, over the whole workgroup (every residue class stopped at its
            // own first one): nothing can happen before it
            {
                const u32 wmin = wave_min_u32_dpp(covered ? 0x7fffffffu : (u32)rr);
                if (lane == 0 && wmin != 0x7fffffffu) atomicMin(reinterpret_cast<unsigned int*>(&shared->clen), wmin);
            }
            __syncthreads();
            const int ru = uni(shared->clen);
            if (ru >= Ev) { PROF_COUNT(25, 1); PROF_MARK(7); break; }
            // ... so the edges up to it only join the adjacency rows, and the next chunk starts AT it: whole chunks of
            // the long-edge tail that hold no candidate are never swept
            if (ru > from) {
                for (int r2 = from + tid; r2 < ru; r2 += NT) {
                    const u32 pk = ord[r2];
                    const int ea = (int)(pk >> 8), eb = (int)(pk & 255u);
                    atomicOr(reinterpret_cast<unsigned long long*>(&adj[2 * ea + (eb >> 6)]), 1ull << (eb & 63));
                    atomicOr(reinterpret_cast<unsigned long long*>(&adj[2 * eb + (ea >> 6)]), 1ull << (ea & 63));
                    // NARROW: no chunk will clear the vectors of the edges that are skipped (ord[r2] is read: dead now)
                    if constexpr (NARROW) { if (r2 < ((psi_bytes - 2 * (E - ru)) >> 2)) psi[r2] = pzero<W, WT>(); }
                }
                PROF_COUNT(35, ru - from);
                clen = ru - r0;                                  // (the loop advances r0 by clen)
                __syncthreads();
            }
        }
        PROF_MARK(7);
        PROF_STOPC(20, out_k0 = k0; out_k1 = k1; out_status = 0);
    }
#ifdef TDA_PROFILE
    PROF_COUNT(36, prof_rneed > 4096 ? 1 : 0); PROF_COUNT(37, prof_rneed > 4608 ? 1 : 0); PROF_COUNT(38, prof_rneed > 5120 ? 1 : 0);
    PROF_COUNT(39, prof_rneed); PROF_COUNT(40, Ev);
#endif
    // essential classes
    const int ncomp = n - merges;
    for (int i = 0; i < ncomp; ++i) {
        if (k0 < h0_cap && tid0 == 0) { h0[2 * k0] = 0.0; h0[2 * k0 + 1] = (double)INFINITY; }
        ++k0;
    }
#pragma unroll
    for (int c = 0; c < W; ++c) {
        u64 al = (u64)alive[c];
        while (al) {
            const int bit = __builtin_ctzll(al);
            al &= al - 1;
            if (k1 < h1_cap && tid0 == 0) { h1[2 * k1] = (double)bkey[WB * c + bit]; h1[2 * k1 + 1] = (double)INFINITY; }
            ++k1;
        }
    }
    if (k1 > h1_cap) status |= TDA_WIN_H1_TRUNCATED;
    out_k0 = k0; out_k1 = k1; out_status = status;
}


// ---------------------------------------------------------------------------------
// The LAST rung of both class ladders: a sweep that cannot run out of anything.  ripser never refuses an input
// (scripts/utils.py:131,140); the chunked sweep above keeps its class vectors in LDS and flags a window whose classes
// do not fit.  This one keeps them in HBM -- one vector of TOT_NW 64-bit words per edge, a bit per class in BIRTH ORDER,
// never reused: a complex on n <= 128 points has at most E - n + 1 <= 8,001 classes, 8,192 bits hold them all -- and
// walks the filtration one edge at a time with the same invariants as the chunked sweep (psi[edge] = class of "edge +
// forest path"; an edge with a common neighbour v* gets psi[a,v*] ^ psi[b,v*]; every other triangle (a,b,v) is
// tested; a non-zero one kills the youngest class in it = its highest bit, which is substituted out of every vector).
// No link argument, no chunks, no capacity: a few barriers per edge, tens of milliseconds per window -- it only ever
// sees the windows every other pass has flagged (lattices, clouds built to have > 64 classes alive at once).
// Vectors that are zero are not stored: one bit per edge in LDS says so.
// ---------------------------------------------------------------------------------
#define TOT_NW 128                    // u64 words per class vector
#define TOT_SLOTS 8                   // windows in flight = workgroups of the launch (8.3 MB of scratch each)
#define TOT_SLOT_WORDS ((size_t)8128 * TOT_NW)

template <int NT, class KEYFN>
__device__ __forceinline__ void rips_sweep_total(int n, int E, int Ev, const u16* rank, const u16* ord, float* bkey, unsigned char* misc,
                                 u64* __restrict__ psi_g, KEYFN keyfn, double* h0, int h0_cap, double* h1, int h1_cap,
                                 int& out_k0, int& out_k1, int& out_status)
{
    constexpr int NWV = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    u64* adj = reinterpret_cast<u64*>(misc + MISC_ADJ);                  // adj[2 v + w]
    int* comp = reinterpret_cast<int*>(misc + MISC_COMP);                // component label of every vertex
    u32* flags = reinterpret_cast<u32*>(misc + MISC_WV);                 // [8] a wave found a non-zero triangle; [8] scalars
    u64* yv = reinterpret_cast<u64*>(misc + MISC_LIST);                  // [TOT_NW] the killing vector
    u64* cur = yv + TOT_NW;                                              // [TOT_NW] psi of the current edge
    u32* nzbit = reinterpret_cast<u32*>(misc + MISC_ADJ + 4096);         // [256] vector of edge r is stored (non-zero)
    u32* alivebit = nzbit + 256;                                         // [256] class c is alive
    static_assert(MISC_LIST_BYTES >= 2 * TOT_NW * 8, "scratch of the total sweep");      // (+ 2 KB behind MISC_BRANK: total_layout)
    for (int i = tid; i < 256; i += NT) { adj[i] = 0ull; nzbit[i] = 0u; alivebit[i] = 0u; }
    for (int v = tid; v < 128; v += NT) comp[v] = v;
    __syncthreads();
    int k0 = 0, k1 = 0, merges = 0, nborn = 0, nalive = 0;
    auto stored = [&](int rr) { return (nzbit[rr >> 5] >> (rr & 31)) & 1u; };
    for (int r = 0; r < Ev; ++r) {
        const u32 pk = (u32)uni((int)ord[r]);
        const int a = (int)(pk >> 8), b = (int)(pk & 255u);
        const u64 M0 = uni64(adj[2 * a] & adj[2 * b]), M1 = uni64(adj[2 * a + 1] & adj[2 * b + 1]);
        if ((M0 | M1) == 0ull) {
            const int ca = uni(comp[a]), cb = uni(comp[b]);
            const float key = keyfn(r, a, b);
            __syncthreads();                                             // (labels read before anybody relabels)
            if (ca != cb) {                                              // negative edge: H0 death
                for (int v = tid; v < n; v += NT) if (comp[v] == cb) comp[v] = ca;
                if (key != 0.0f) {
                    if (tid == 0 && k0 < h0_cap) { h0[2 * k0] = 0.0; h0[2 * k0 + 1] = (double)key; }
                    ++k0;
                }
                ++merges;
            } else {                                                     // a class is born: the next bit
                const int c = nborn++;
                if (tid == 0) { bkey[c] = key; nzbit[r >> 5] |= 1u << (r & 31); alivebit[c >> 5] |= 1u << (c & 31); }
                for (int w = tid; w < TOT_NW; w += NT) psi_g[(size_t)r * TOT_NW + w] = (w == (c >> 6)) ? 1ull << (c & 63) : 0ull;
                ++nalive;
            }
        } else if (nalive > 0) {
            const int vstar = M0 ? __builtin_ctzll(M0) : 64 + __builtin_ctzll(M1);
            const int s1 = uni((int)rank[pair_index(a, vstar)]), s2 = uni((int)rank[pair_index(b, vstar)]);
            const bool z1 = !stored(s1), z2 = !stored(s2);
            u64 bw = 0ull;
            if (tid < TOT_NW) {
                bw = (z1 ? 0ull : psi_g[(size_t)s1 * TOT_NW + tid]) ^ (z2 ? 0ull : psi_g[(size_t)s2 * TOT_NW + tid]);
                cur[tid] = bw;
            }
            if (lane == 0) flags[wave] = 0u;
            __syncthreads();
            if (__ballot(bw != 0ull) != 0ull && lane == 0) flags[wave] = 1u;
            __syncthreads();
            bool nzr = false;
#pragma unroll
            for (int w = 0; w < NWV; ++w) nzr = nzr || flags[w] != 0u;
            if (nzr) {
                if (tid < TOT_NW) psi_g[(size_t)r * TOT_NW + tid] = bw;
                if (tid == 0) nzbit[r >> 5] |= 1u << (r & 31);
            }
            // the other triangles, NWV at a time (one per wave); a kill changes the table: the same ones are looked at again
            u64 R0 = M0, R1 = M1;
            if (vstar < 64) R0 &= ~(1ull << vstar); else R1 &= ~(1ull << (vstar - 64));
            float keyr = 0.f;
            bool have_key = false;
            while ((R0 | R1) != 0ull && nalive > 0) {
                __syncthreads();                                         // flags / cur of the round before are done with
                // wave w takes the w-th remaining common neighbour
                int v = -1;
                {
                    u64 t0 = R0, t1 = R1;
                    for (int k = 0; k < wave && (t0 | t1); ++k) { if (t0) t0 &= t0 - 1ull; else t1 &= t1 - 1ull; }
                    if (t0) v = __builtin_ctzll(t0); else if (t1) v = 64 + __builtin_ctzll(t1);
                }
                u64 y0 = 0ull, y1 = 0ull;
                if (v >= 0) {
                    const int ja = uni((int)rank[pair_index(a, v)]), jb = uni((int)rank[pair_index(b, v)]);
                    y0 = cur[lane]; y1 = cur[lane + 64];
                    if (stored(ja)) { y0 ^= psi_g[(size_t)ja * TOT_NW + lane]; y1 ^= psi_g[(size_t)ja * TOT_NW + lane + 64]; }
                    if (stored(jb)) { y0 ^= psi_g[(size_t)jb * TOT_NW + lane]; y1 ^= psi_g[(size_t)jb * TOT_NW + lane + 64]; }
                }
                const bool wnz = __ballot((y0 | y1) != 0ull) != 0ull;
                if (lane == 0) flags[wave] = wnz ? 1u : 0u;
                __syncthreads();
                int first = -1;
#pragma unroll
                for (int w = NWV - 1; w >= 0; --w) if (flags[w]) first = w;
                if (first < 0) {                                         // these NWV triangles are trivial: on to the next ones
                    for (int k = 0; k < NWV && (R0 | R1); ++k) { if (R0) R0 &= R0 - 1ull; else R1 &= R1 - 1ull; }
                    continue;
                }
                if (wave == first) {                                     // the killer: publish the vector and its youngest class
                    yv[lane] = y0; yv[lane + 64] = y1;
                    int m = y1 ? 64 * (lane + 64) + 63 - __builtin_clzll(y1) : (y0 ? 64 * lane + 63 - __builtin_clzll(y0) : -1);
                    m = wave_max_i32_dpp(m);
                    if (lane == 0) flags[8] = (u32)m;
                }
                __syncthreads();
                const int c = (int)flags[8];
                if (!have_key) { keyr = keyfn(r, a, b); have_key = true; }
                const float birth = bkey[c];
                if (keyr > birth) {
                    if (tid == 0 && k1 < h1_cap) { h1[2 * k1] = (double)birth; h1[2 * k1 + 1] = (double)keyr; }
                    ++k1;
                }
                if (tid == 0) alivebit[c >> 5] &= ~(1u << (c & 31));
                --nalive;
                // substitute the class out of every stored vector up to this edge (one vector per wave and trip)
                for (int rr = wave; rr <= r; rr += NWV) {
                    if (!stored(rr)) continue;
                    const u64 wc = psi_g[(size_t)rr * TOT_NW + (c >> 6)];
                    if (!((wc >> (c & 63)) & 1ull)) continue;
                    psi_g[(size_t)rr * TOT_NW + lane] ^= yv[lane];
                    psi_g[(size_t)rr * TOT_NW + lane + 64] ^= yv[lane + 64];
                }
                __syncthreads();                                         // (global stores of a workgroup are visible to it behind a barrier)
                if (tid < TOT_NW) cur[tid] = stored(r) ? psi_g[(size_t)r * TOT_NW + tid] : 0ull;
            }
        }
        __syncthreads();
        if (tid == 0) { adj[2 * a + (b >> 6)] |= 1ull << (b & 63); adj[2 * b + (a >> 6)] |= 1ull << (a & 63); }
        __syncthreads();
    }
    int status = 0;
    const int ncomp = n - merges;
    for (int i = 0; i < ncomp; ++i) {
        if (k0 < h0_cap && tid == 0) { h0[2 * k0] = 0.0; h0[2 * k0 + 1] = (double)INFINITY; }
        ++k0;
    }
    for (int c = 0; c < nborn; ++c)
        if ((alivebit[c >> 5] >> (c & 31)) & 1u) {
            if (k1 < h1_cap && tid == 0) { h1[2 * k1] = (double)bkey[c]; h1[2 * k1 + 1] = (double)INFINITY; }
            ++k1;
        }
    if (k1 > h1_cap) status |= TDA_WIN_H1_TRUNCATED;
    out_k0 = k0; out_k1 = k1; out_status = status;
}

// ---------------------------------------------------------------------------------
// distance-matrix flavour (EEG): LDS = [S | psi] [ord] [rank] [skey] [misc]
// ---------------------------------------------------------------------------------
struct KeyFromLds {
    const u32* skey;
    __device__ __forceinline__ float operator()(int r, int, int) const { return sortable_f32(skey[r]); }
};

// everything after the keys: key32[e] (flat edge order) and vmax[v] = max_u key(v,u) are in LDS, barrier passed
template <int NT, int NVW, int W, typename WT, bool TOTAL = false>
__device__ __forceinline__ void rips_dm_rest(unsigned char* smem, const int win, int n, float thresh, const u32* vmax, u32 kmin_thread,
                             const RipsLayout& L, const RipsOut& out, u64* psi_g = nullptr)
{
    const int tid = threadIdx.x;
    const int E = tri2(n);
    u32* key32 = reinterpret_cast<u32*>(smem);
    u16* members = reinterpret_cast<u16*>(smem + L.off_members);
    Psi<W, WT>* psi = reinterpret_cast<Psi<W, WT>*>(smem);
    u16* rank = reinterpret_cast<u16*>(smem + L.off_rank);
    u16* ord = reinterpret_cast<u16*>(smem + L.off_ord);
    u32* skey = reinterpret_cast<u32*>(smem + L.off_aux);
    unsigned char* misc = smem + L.off_misc;
    u32* red = reinterpret_cast<u32*>(misc + MISC_MIN);
    u32* cursor = reinterpret_cast<u32*>(misc + MISC_SORTCNT);
    int* wsum = reinterpret_cast<int*>(misc + MISC_WV);
    const u32 tkey = f32_sortable(thresh);
    u32 teff, kmin;
    PROF_RESUME();
    effective_threshold<NT>(n, tkey, vmax, kmin_thread, red, teff, kmin);
    PROF_STOP(1, if (tid == 0) { out.h0_cnt[win] = 0; out.h1_cnt[win] = 0; out.status[win] = 0; });
    const int Ev = rank_edges<NT, (NVW == 1 ? 2048 : 8192), true>(key32, E, teff, kmin, members, cursor, wsum, rank, ord, skey);
    PROF_MARK(1);
    PROF_STOP(2, if (tid == 0) { out.h0_cnt[win] = 0; out.h1_cnt[win] = 0; out.status[win] = 0; });
    guard_write(smem, L);
    int k0, k1, st;
    KeyFromLds kf{skey};
    if constexpr (TOTAL)
        rips_sweep_total<NT>(n, E, Ev, rank, ord, reinterpret_cast<float*>(smem), misc, psi_g, kf,
                             out.h0 + (size_t)win * out.h0_cap * 2, out.h0_cap,
                             out.h1 + (size_t)win * out.h1_cap * 2, out.h1_cap, k0, k1, st);
    else
    rips_sweep<NT, NVW, W, WT, false>(n, E, Ev, rank, ord, psi, 0, misc, kf,
                       out.h0 + (size_t)win * out.h0_cap * 2, out.h0_cap,
                       out.h1 + (size_t)win * out.h1_cap * 2, out.h1_cap, k0, k1, st);
    st |= guard_check(smem, L);
    PROF_MARK(3);
    PROF_COUNT(8, 1);
    PROF_COUNT(9, E);
    PROF_FLUSH();
    if (tid == 0) { out.h0_cnt[win] = k0; out.h1_cnt[win] = k1; out.status[win] = st; }
}

template <int NT, int NVW, int W, typename WT, bool TOTAL = false>
__device__ __forceinline__ void rips_dm_window(unsigned char* smem, const int win, const double* __restrict__ dm, int n, float thresh,
                               int symmetrise, const RipsLayout& L, const RipsOut& out, u64* psi_g = nullptr)
{
    const int tid = threadIdx.x;
    const int E = tri2(n);
    u32* key32 = reinterpret_cast<u32*>(smem);
    unsigned char* misc = smem + L.off_misc;
    PROF_BEGIN();
    const double* D = dm + (size_t)win * n * n;
    // P0. keys: utils.py:137-139 then ripser's float32 cast
    u32* vmax = reinterpret_cast<u32*>(misc + MISC_COMP);
    if (tid < 128) vmax[tid] = 0u;
    __syncthreads();
    u32 kmin_thread = 0xffffffffu;
    for (int e = tid; e < E; e += NT) {
        const int a = edge_row(e), b = e - tri2(a);
        double v;
        if (symmetrise) {
            v = (D[(size_t)a * n + b] + D[(size_t)b * n + a]) / 2.0;
            if (v < 0.0) v = 0.0;
        } else {
            v = D[(size_t)b * n + a];
        }
        const u32 sk = f32_sortable((float)v);
        key32[e] = sk;
        kmin_thread = sk < kmin_thread ? sk : kmin_thread;
        atomicMax(&vmax[a], sk);
        atomicMax(&vmax[b], sk);
    }
    __syncthreads();
    PROF_MARK(0);
    rips_dm_rest<NT, NVW, W, WT, TOTAL>(smem, win, n, thresh, vmax, kmin_thread, L, out, psi_g);
}

#ifdef TDA_DEBUG_PTS
__device__ double g_dbg[2048];
extern "C" __attribute__((visibility("default"))) int tda_debug_read(double* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(double) * 2048) != hipSuccess;
}
#endif
// ---------------------------------------------------------------------------------
// point-cloud flavour (audio): LDS = [S | psi] [ord] [rank] [pts f64] [misc]
// mode 0: Takens embedding of a window (utils.py:107-116) + min-max (utils.py:127-130)
// mode 1: explicit (P, dim) cloud, optional min-max
// ---------------------------------------------------------------------------------
struct KeyFromPts {
    const double* pts;   // normalised cloud, (P, dim)
    int dim;
    __device__ __forceinline__ float operator()(int, int a, int b) const
    {
        // sklearn euclidean_distances: -2 x.y + |x|^2 + |y|^2, clamp, sqrt  (see oracle)
        double na = 0.0, nb = 0.0, dot = 0.0;
        if (dim == 3) {                          // the reference's TAKENS_DIM: the same operations, unrolled
            const double* pa = pts + 3 * a;
            const double* pb = pts + 3 * b;
            const double a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
            na = (a0 * a0 + a1 * a1) + a2 * a2;  // (0.0 + a0^2 is a0^2 exactly)
            nb = (b0 * b0 + b1 * b1) + b2 * b2;
            dot = fma(a2, b2, fma(a1, b1, a0 * b0));
        } else
        for (int k = 0; k < dim; ++k) {
            const double xa = pts[a * dim + k], xb = pts[b * dim + k];
            na += xa * xa;
            nb += xb * xb;
            dot = (k == 0) ? xa * xb : fma(xa, xb, dot);
        }
        double d2 = -2.0 * dot;
        d2 += na;
        d2 += nb;
        if (!(d2 > 0.0)) d2 = 0.0;
        return (float)sqrt(d2);
    }
};

// The same with |x|^2 of every point from a table (the key pass: every point takes part in P - 1 edges; the table holds
// exactly the value the expression above computes, so the keys are bit-identical)
struct KeyFromPtsTab {
    const double* pts; const double* nrm; int dim;
    __device__ __forceinline__ static double norm2(const double* p, int dim)
    {
        if (dim == 3) return (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
        double n = 0.0;
        for (int k = 0; k < dim; ++k) n += p[k] * p[k];
        return n;
    }
    __device__ __forceinline__ float operator()(int, int a, int b) const
    {
        double dot = 0.0;
        if (dim == 3) {
            const double* pa = pts + 3 * a;
            const double* pb = pts + 3 * b;
            dot = fma(pa[2], pb[2], fma(pa[1], pb[1], pa[0] * pb[0]));
        } else
        for (int k = 0; k < dim; ++k) {
            const double xa = pts[a * dim + k], xb = pts[b * dim + k];
            dot = (k == 0) ? xa * xb : fma(xa, xb, dot);
        }
        double d2 = -2.0 * dot;
        d2 += nrm[a];
        d2 += nrm[b];
        if (!(d2 > 0.0)) d2 = 0.0;
        return (float)sqrt_rn(d2);
    }
};

// The row part of max_u d(v,u) for a cloud: the lanes of a wave sit in one or two rows, so an atomic per edge would
// hit one address 64 times.  NT / 128 threads per vertex walk the finished row instead, a slice each.  (Not inlined:
// the point-cloud kernel sits at its register limit and the inlined loop tipped it into spilling.)
template <int NT>
__device__ __forceinline__ void row_maxima(const u32* key32, u32* vmax, int P)
{
    constexpr int TPV = NT / 128;
    const int tid = threadIdx.x;
    const int v = tid / TPV, part = tid - v * TPV;
    if (v < P) {
        const int t0 = tri2(v);
        u32 m = 0u;
        for (int b = part; b < v; b += TPV) { const u32 k = key32[t0 + b]; m = k > m ? k : m; }
        if (m) atomicMax(&vmax[v], m);
    }
}

// Widening passes: the windows the pass before left flagged are collected into a list first (retry_collect_kernel: one
// coalesced look at every status word, a wave-aggregated append), and workgroup b of the pass redoes entries b,
// b + gridDim.x, ...: every workgroup gets its share whatever the order of the flags.  (The passes used to walk the
// status array themselves, NT consecutive words per workgroup: a batch of 17,700 windows kept 35 workgroups busy with ten
// flagged windows each, one after the other -- 1.2 ms per pass on a nearly empty chip.)  A pass with nothing to redo
// costs one load per workgroup.
__global__ void __launch_bounds__(256) retry_collect_kernel(const int* __restrict__ status, int n_win, int* __restrict__ list)
{
    const int w = blockIdx.x * 256 + (int)threadIdx.x;
    const bool f = w < n_win && (status[w] & TDA_WIN_CLASS_OVERFLOW);
    const u64 bal = __ballot(f);
    if (bal) {
        const int lane = (int)threadIdx.x & 63, lead = __builtin_ctzll(bal);
        int base = 0;
        if (lane == lead) base = atomicAdd(&list[0], __builtin_popcountll(bal));
        base = __builtin_amdgcn_readlane(base, lead);
        if (f) list[4 + base + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = w;
    }
}
// The pass empties the list when it is through with it ([1] counts the workgroups that are: the last one clears both
// words), so that the next collection starts from zero without a memset in between.
#define RETRY_SCAN_BEGIN(NT_, out_, n_win_)                                                        \
    {                                                                                              \
        int* rl__ = (out_).retry_list;                                                             \
        const int nl__ = uni(rl__[0]);                                                             \
        for (int j__ = blockIdx.x; j__ < nl__; j__ += gridDim.x) {                                 \
            const int win = uni(rl__[4 + j__]);
#define RETRY_SCAN_END()                                                                           \
            __syncthreads();                                                                       \
        }                                                                                          \
        if (threadIdx.x == 0 && atomicAdd(&rl__[1], 1) == (int)gridDim.x - 1) {                    \
            rl__[0] = 0;                                                                           \
            rl__[1] = 0;                                                                           \
        }                                                                                          \
    }

// Kernel shell shared by both flavours.  First pass: one workgroup per window.  Retry passes (wider
// class vector) run a grid that fits the chip and redo the windows on the list of the flagged ones
// (RETRY_SCAN_BEGIN), so a retry with nothing to do costs a few microseconds instead of n_win LDS-heavy
// workgroup launches.
// 64 and 128 classes: four waves per SIMD (128 VGPRs, no spills).  Five (96 VGPRs, 32 B of scratch per lane) paid while
// the sweep spent its time at barriers; with the one-barrier votes four is 1 % faster end to end
template <int NT, int NVW, int W, typename WT, bool RETRY>
__global__ void __launch_bounds__(NT, (W <= 2 && !RETRY) ? 4 : 2)
rips_dm_kernel(const double* __restrict__ dm, int n_win, int n, float thresh, int symmetrise, RipsLayout L,
               RipsOut out, unsigned long long* __restrict__ retry_ctr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (!RETRY) {                          // one window per workgroup, no loop (see eeg_window_kernel)
        if ((int)blockIdx.x < n_win)
            rips_dm_window<NT, NVW, W, WT>(smem, (int)blockIdx.x, dm, n, thresh, symmetrise, L, out);
    } else {
        RETRY_SCAN_BEGIN(NT, out, n_win)
            if (retry_ctr && threadIdx.x == 0) atomicAdd(retry_ctr, 1ull);
            rips_dm_window<NT, NVW, W, WT>(smem, win, dm, n, thresh, symmetrise, L, out);
        RETRY_SCAN_END()
    }
}

template <int NT, int W, typename WT, bool NARROW, bool TOTAL = false>
__device__ __forceinline__ void rips_cloud_window(unsigned char* smem, const int win, const double* __restrict__ src,
                                  const int* __restrict__ tau_or_npts, int n_t_or_pcap, int dim, int subsample,
                                  int mode, int normalise, float thresh, const RipsLayout& L, int p_max,
                                  int* __restrict__ n_points, const RipsOut& out, u64* psi_g = nullptr)
{
    const int tid = threadIdx.x;
    u32* key32 = reinterpret_cast<u32*>(smem);
    u16* members = reinterpret_cast<u16*>(smem + (NARROW ? NL::MEMBERS : L.off_members));
    Psi<W, WT>* psi = reinterpret_cast<Psi<W, WT>*>(smem + (NARROW ? NL::PSI : 0));
    u16* rank = reinterpret_cast<u16*>(smem + (NARROW ? NL::RANK : L.off_rank));
    u16* ord = reinterpret_cast<u16*>(smem + L.off_ord);         // NARROW: at the end of the region of the class vectors (below)
    double* pts = reinterpret_cast<double*>(smem + (NARROW ? NL::AUX : L.off_aux));
    unsigned char* misc = smem + (NARROW ? NL::MISC : L.off_misc);
    // scratch of the phases before the sweep: the head of `misc`; the narrow layout keeps it clear of the ranking arrays
    unsigned char* rscr = smem + (NARROW ? NL::RSCR : L.off_misc);
    u32* red = reinterpret_cast<u32*>(rscr + MISC_MIN);
    double* mm = reinterpret_cast<double*>(rscr + (NARROW ? MISC_CAND : MISC_CKEY));    // min/max scratch

    double* h0 = out.h0 + (size_t)win * out.h0_cap * 2;
    double* h1 = out.h1 + (size_t)win * out.h1_cap * 2;

    int P;
    const double* base;
    int tau = 0;
    if (mode == 0) {
        tau = tau_or_npts[win];
        const int nn = n_t_or_pcap - (dim - 1) * tau;
        P = nn > 0 ? (nn + subsample - 1) / subsample : 0;
        base = src + (size_t)win * n_t_or_pcap;
    } else {
        P = tau_or_npts[win];
        base = src + (size_t)win * n_t_or_pcap * dim;
    }
    if (n_points && tid == 0) n_points[win] = P;
    if (P > p_max || (mode == 0 && tau < 1)) {
        if (tid == 0) { out.h0_cnt[win] = 0; out.h1_cnt[win] = 0; out.status[win] = TDA_WIN_TOO_LARGE; }
        return;
    }
    if (P < 3) {
        // utils.py:125-126: [[0,0]], [[0,0]]
        if (tid == 0) {
            if (out.h0_cap > 0) { h0[0] = 0.0; h0[1] = 0.0; }
            if (out.h1_cap > 0) { h1[0] = 0.0; h1[1] = 0.0; }
            out.h0_cnt[win] = 1; out.h1_cnt[win] = 1;
            out.status[win] = TDA_WIN_DEGENERATE;
        }
        return;
    }
    PROF_BEGIN();
    // cloud -> LDS, per-column min-max to [0,1] (range 0 -> 1); P <= 128: one wave per column pass
    for (int idx = tid; idx < P * dim; idx += NT) {
        const int i = idx / dim, k = idx - i * dim;
        pts[idx] = (mode == 0) ? base[i * subsample + k * tau] : base[idx];
    }
    __syncthreads();
    if (normalise) {
        if ((tid >> 6) < dim) {                            // one wave per column (dim <= 4)
            const int k = tid >> 6, l = tid & 63;
            double mn = INFINITY, mx = -INFINITY;
            for (int i = l; i < P; i += 64) {
                const double v = pts[i * dim + k];
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
            }
            mn = wave_min_f64(mn);
            mx = wave_max_f64(mx);
            if (l == 0) { double rg = mx - mn; if (rg == 0.0) rg = 1.0; mm[2 * k] = mn; mm[2 * k + 1] = rg; }
        }
        __syncthreads();
        for (int idx = tid; idx < P * dim; idx += NT) {
            const int k = idx % dim;
            pts[idx] = (pts[idx] - mm[2 * k]) / mm[2 * k + 1];
        }
        __syncthreads();
    }
#ifdef TDA_DEBUG_PTS
    for (int idx = tid; idx < P * dim; idx += NT) g_dbg[idx] = pts[idx];
#endif
    // P0. keys
    const int E = tri2(P);
    const u32 tkey = f32_sortable(thresh);
    KeyFromPts kf{pts, dim};
    u32* vmax = reinterpret_cast<u32*>(rscr + MISC_COMP);
    u32* cursor = reinterpret_cast<u32*>(smem + (NARROW ? NL::CURSOR : L.off_cursor));
    int* wsum = reinterpret_cast<int*>(rscr + MISC_WV);
    if (tid < 128) vmax[tid] = 0u;
    // |x|^2 of every point, once (in the bucket cursors' place: they are not in use before the ranking)
    double* nrm = reinterpret_cast<double*>(cursor);
    if (tid < P) nrm[tid] = KeyFromPtsTab::norm2(pts + tid * dim, dim);
    __syncthreads();
    const KeyFromPtsTab kft{pts, nrm, dim};
    u32 kmin_thread = 0xffffffffu;
    u32 ab_next = edge_ab(tid);                 // (the table is padded: one trip ahead without a guard)
    for (int e = tid; e < E; e += NT) {
        const u32 ab = ab_next;
        ab_next = edge_ab(e + NT);
        const int a = (int)(ab >> 8), b = (int)(ab & 255u);
        const u32 sk = f32_sortable(kft(0, a, b));
        key32[e] = sk;
        kmin_thread = sk < kmin_thread ? sk : kmin_thread;
        atomicMax(&vmax[b], sk);          // (consecutive lanes: consecutive b, no conflict)
    }
    __syncthreads();
    row_maxima<NT>(key32, vmax, P);
    __syncthreads();
    u32 teff, kmin;
    effective_threshold<NT>(P, tkey, vmax, kmin_thread, red, teff, kmin);
    PROF_MARK(0);
    PROF_STOP(1, if (tid == 0) { out.h0_cnt[win] = 0; out.h1_cnt[win] = 0; out.status[win] = 0; });
    int Ev;
    if constexpr (NARROW) {
        ord = reinterpret_cast<u16*>(smem + NL::PSI + NL::REGION) - E;        // ord[E - 1] ends the region
        Ev = rank_edges_narrow<NT, NARROW_NB>(key32, E, teff, kmin, members, cursor, wsum, rank, ord);
    }
    else Ev = rank_edges<NT, CLOUD_NB, false>(key32, E, teff, kmin, members, cursor, wsum, rank, ord, nullptr);
    PROF_MARK(1);
    PROF_STOP(2, if (tid == 0) { out.h0_cnt[win] = 0; out.h1_cnt[win] = 0; out.status[win] = 0; });
    guard_write(smem, L);
    int k0, k1, st;
    if constexpr (TOTAL)
        rips_sweep_total<NT>(P, E, Ev, rank, ord, reinterpret_cast<float*>(smem), misc, psi_g, kf, h0, out.h0_cap, h1, out.h1_cap, k0, k1, st);
    else if (P <= 64)
        rips_sweep<NT, 1, W, WT, NARROW>(P, E, Ev, rank, ord, psi, NL::REGION, misc, kf, h0, out.h0_cap, h1, out.h1_cap, k0, k1, st);
    else
        rips_sweep<NT, 2, W, WT, NARROW>(P, E, Ev, rank, ord, psi, NL::REGION, misc, kf, h0, out.h0_cap, h1, out.h1_cap, k0, k1, st);
    st |= guard_check(smem, L);
    PROF_MARK(3);
    PROF_COUNT(8, 1);
    PROF_COUNT(9, E);
    PROF_FLUSH();
#ifdef TDA_DEBUG_PTS
    __syncthreads();
    for (int idx = tid; idx < P * dim; idx += NT) g_dbg[1024 + idx] = pts[idx];
#endif
    if (tid == 0) { out.h0_cnt[win] = k0; out.h1_cnt[win] = k1; out.status[win] = st; }
}

// two 512-thread workgroups per CU = 4 waves per SIMD (second launch-bound parameter of hip-clang): 128 VGPRs
#ifndef CLOUD_WAVES
#define CLOUD_WAVES 4
#endif
// NARROW: three workgroups of 384 threads per CU = 18 waves, five per SIMD at most: 96 VGPRs.  (Three workgroups of 512
// would need six waves per SIMD: at 80 VGPRs the sweep spills the thread index and half of its scalars, and every
// phase, the single-wave stretches included, ran 20-40 % longer -- measured: more than the third workgroup brought.)
template <int NT, int W, typename WT, bool RETRY, bool NARROW = false>
__global__ void __launch_bounds__(NT, RETRY ? 2 : (NARROW ? NARROW_WAVES : CLOUD_WAVES))
rips_cloud_kernel(const double* __restrict__ src, const int* __restrict__ tau_or_npts, int n_win,
                  int n_t_or_pcap, int dim, int subsample, int mode, int normalise, float thresh,
                  RipsLayout L, int p_max, int* __restrict__ n_points, RipsOut out,
                  unsigned long long* __restrict__ span, unsigned long long* __restrict__ retry_ctr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // armed probe only (100 MHz wall clock): span[0] = start of workgroup 0 of this launch, span[1] = workgroups
    // done; the last workgroup adds (its end - that start), the interval a kernel trace reports, to span[2], counts
    // the launch in span[3] and re-arms the counter, so that a launch baked into a HIP graph keeps measuring.  (One
    // store at the start: 512 simultaneous atomics on one address cost ~50 us per launch.)
    if (span && blockIdx.x == 0 && threadIdx.x == 0)      // workgroups are dispatched in order: 0 starts first
        atomicExch(&span[0], wall_clock64());
    if constexpr (!RETRY) {                               // one window per workgroup, no loop (see eeg_window_kernel)
        if ((int)blockIdx.x < n_win)
            rips_cloud_window<NT, W, WT, NARROW>(smem, (int)blockIdx.x, src, tau_or_npts, n_t_or_pcap, dim, subsample, mode,
                                                 normalise, thresh, L, p_max, n_points, out);
    } else {
        RETRY_SCAN_BEGIN(NT, out, n_win)
            if (retry_ctr && threadIdx.x == 0) atomicAdd(retry_ctr + 1, 1ull);
            rips_cloud_window<NT, W, WT, false>(smem, win, src, tau_or_npts, n_t_or_pcap, dim, subsample, mode, normalise,
                                                thresh, L, p_max, n_points, out);
        RETRY_SCAN_END()
    }
    if (span && threadIdx.x == 0) {                        // device-scope atomics only: a fence would write L2 back
        const unsigned long long now = wall_clock64();
        if (atomicAdd(&span[1], 1ull) == (unsigned long long)gridDim.x - 1ull) {
            const unsigned long long first = atomicAdd(&span[0], 0ull);
            atomicAdd(&span[2], now - first);
            atomicAdd(&span[3], 1ull);
            atomicExch(&span[1], 0ull);
        }
    }
}

// The last rung (rips_sweep_total): TOT_SLOTS workgroups walk the status words and redo what is still flagged, each with
// its own 8.3 MB of HBM for the class vectors.
template <int NT>
__global__ void __launch_bounds__(NT, 2)
rips_dm_total_kernel(const double* __restrict__ dm, int n_win, int n, float thresh, int symmetrise, RipsLayout L,
                     RipsOut out, u64* __restrict__ scratch, unsigned long long* __restrict__ retry_ctr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* psi_g = scratch + (size_t)blockIdx.x * TOT_SLOT_WORDS;
    RETRY_SCAN_BEGIN(NT, out, n_win)
        if (retry_ctr && threadIdx.x == 0) atomicAdd(retry_ctr + 2, 1ull);
        rips_dm_window<NT, 2, 1, u32, true>(smem, win, dm, n, thresh, symmetrise, L, out, psi_g);
    RETRY_SCAN_END()
}
template <int NT>
__global__ void __launch_bounds__(NT, 2)
rips_cloud_total_kernel(const double* __restrict__ src, const int* __restrict__ tau_or_npts, int n_win, int n_t_or_pcap,
                        int dim, int subsample, int mode, int normalise, float thresh, RipsLayout L, int p_max,
                        int* __restrict__ n_points, RipsOut out, u64* __restrict__ scratch,
                        unsigned long long* __restrict__ retry_ctr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* psi_g = scratch + (size_t)blockIdx.x * TOT_SLOT_WORDS;
    RETRY_SCAN_BEGIN(NT, out, n_win)
        if (retry_ctr && threadIdx.x == 0) atomicAdd(retry_ctr + 2, 1ull);
        rips_cloud_window<NT, 1, u32, false, true>(smem, win, src, tau_or_npts, n_t_or_pcap, dim, subsample, mode, normalise,
                                                   thresh, L, p_max, n_points, out, psi_g);
    RETRY_SCAN_END()
}

// ---------------------------------------------------------------------------------
// Fused EEG window kernel: window (n_ch x n_t float64) -> correlation -> distance -> float32 keys -> Rips H0/H1, one
// launch, one window per 256-thread workgroup; the distance matrix never leaves LDS (95.1 KB of algorithmic HBM
// traffic per window instead of 129.4: SURVEY.md section 8d).  Replaces, per window, compute_correlation_matrix +
// correlation_to_distance (notebooks/2_graph_construction.ipynb:86-122) followed by compute_eeg_persistence
// (scripts/utils.py:135-141): the keys are float32((D[a][b] + D[b][a]) / 2) with D[i][j] = sqrt(2 (1 - r_ij)),
// r_ij = (cov_ij / s_i) / s_j -- the arithmetic of corr_dist_kernel and of the key stage of rips_dm_kernel, operation
// for operation, so the diagrams are bit-identical to the two-kernel path.  dist / corr (optional) receive the
// matrices as corr_dist_kernel writes them.
// ---------------------------------------------------------------------------------
// where the windows of a launch live: stacked (preprocessed/<rec>/<band>.npy: win_stride = n_ch * n_t, ld = n_t, one
// group) or sliding over band-passed recordings of equal length (n_rec, n_ch, n_samples): window k of recording r
// starts at r * group_stride + k * win_stride, row stride ld = n_samples -- create_sliding_windows
// (notebooks/1_preprocesamiento.ipynb:314-381) without the 4x overlapping stack.  sel (optional): the windows to
// process, as indices r * wins_per_group + k (the window selection of the drivers).
struct WindowSource {
    const double* base; long long win_stride, group_stride; int ld, wins_per_group; const int* sel;
    __device__ __forceinline__ const double* at(int w) const
    {
        const int g = sel ? sel[w] : w;
        const int r = g / wins_per_group, k = g - r * wins_per_group;
        return base + (size_t)r * (size_t)group_stride + (size_t)k * (size_t)win_stride;
    }
};

template <int NB, bool RES, int W, typename WT>
__device__ __forceinline__ void eeg_one_window(unsigned char* smem, const WindowSource& src, int w, int n_ch,
                                               int n_t, float thresh, const RipsLayout& L, const RipsOut& out,
                                               double* __restrict__ dist, double* __restrict__ corr)
{
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
    PROF_BEGIN();
    cd_window_products<NB, RES>(smem, src.at(w), n_ch, n_t, src.ld);
    const double* accs = reinterpret_cast<const double*>(smem) + CdLayout<NB>::TILE;
    const double* sdev = accs + CdLayout<NB>::ACCS + CdLayout<NB>::CP;
    const double fact = 1.0 / (double)(n_t - 1);
    auto dist_at = [&](int i, int j, double* r_out) {
        double r = (accs[CdLayout<NB>::acc_index(i, j)] * fact / sdev[i]) / sdev[j];
        r = r > 1.0 ? 1.0 : r; r = r < -1.0 ? -1.0 : r; if (r != r) r = 0.0;       // clip; nan_to_num (nb2:95)
        if (r_out) *r_out = r;
        double d = sqrt_rn(2.0 * (1.0 - r));                                     // nb2:108 (sqrt() bit for bit: common.h)
        if (!(d > 0.0) || i == j) d = 0.0;                                       // nb2:119-120
        return d;
    };
    // keys into the (free) tile area; the row maxima next to them, in what becomes the members array
    const int E = tri2(n_ch);
    u32* key32 = reinterpret_cast<u32*>(smem);
    u32* vmax = reinterpret_cast<u32*>(smem + 4 * ((E + 3) & ~3));
    if (tid < 128) vmax[tid] = 0u;
    __syncthreads();
    u32 kmin_thread = 0xffffffffu;
    for (int e = tid; e < E; e += 256) {
        const int a = edge_row(e), b = e - tri2(a);
        double v = (dist_at(a, b, nullptr) + dist_at(b, a, nullptr)) / 2.0;      // utils.py:137
        if (v < 0.0) v = 0.0;                                                    // utils.py:139
        const u32 sk = f32_sortable((float)v);
        key32[e] = sk;
        kmin_thread = sk < kmin_thread ? sk : kmin_thread;
        atomicMax(&vmax[a], sk);
        atomicMax(&vmax[b], sk);
    }
    if (dist) {                                  // the matrices themselves, when the caller wants them on disk
        double* Dw = dist + (size_t)w * n_ch * n_ch;
        double* Cw = corr ? corr + (size_t)w * n_ch * n_ch : nullptr;
        for (int i = wv; i < n_ch; i += 4)
            if (l < n_ch) {
                double r;
                const double d = dist_at(i, l, &r);
                Dw[i * n_ch + l] = d;
                if (Cw) Cw[i * n_ch + l] = r;
            }
    }
    __syncthreads();
    PROF_MARK(0);
    rips_dm_rest<256, 1, W, WT>(smem, w, n_ch, thresh, vmax, kmin_thread, L, out);
}

// RETRY = false: one window per workgroup and NO loop over windows -- with the loop the compiler hoists the address
// arithmetic of the window fetch out of it and the kernel spills 141 registers (measured).
// Launch bound: never ONE wave per SIMD -- at that bound hipcc keeps the f64 MFMA accumulators of cd_window_products in
// AGPRs and mis-places the wait in front of the first v_accvgpr_read of one path (a proven compiler fault: see the
// note in corr_dist_dev.h; the sweep and its LDS hand-overs are not involved -- the distance matrix itself comes out
// wrong).  Every variant of this kernel stays at >= 2 waves per SIMD (no AGPRs: checked by a CPU test on the code
// object) and tests/test_gpu_parity.py::test_fused_eeg_window_512_classes covers the widest one.
#ifndef TDA_EEG_WIDE_WAVES
#define TDA_EEG_WIDE_WAVES 2     // (tools/probes/wide_waves_repro.py builds with 1 to reproduce the fault)
#endif
template <int NB, bool RES, int W, bool RETRY, typename WT>
// first pass of the fused kernel: six workgroups per CU (80 VGPRs, 24 B of scratch per lane; 26.9 KB of LDS with
// 32-sample tiles).  Four (111 VGPRs, 38 KB with 64-sample tiles) left half the issue slots empty: +13 % on the kernel
#ifndef TDA_EEG_WAVES
#define TDA_EEG_WAVES 6
#endif
__global__ void __launch_bounds__(256, (W > 2 || RETRY) ? TDA_EEG_WIDE_WAVES : (RES ? 3 : TDA_EEG_WAVES))
eeg_window_kernel(WindowSource windows, int n_win, int n_ch, int n_t, float thresh, RipsLayout L, RipsOut out,
                  double* __restrict__ dist, double* __restrict__ corr, unsigned long long* __restrict__ retry_ctr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (!RETRY) {
        if ((int)blockIdx.x < n_win)
            eeg_one_window<NB, RES, W, WT>(smem, windows, (int)blockIdx.x, n_ch, n_t, thresh, L, out, dist, corr);
    } else {
        RETRY_SCAN_BEGIN(256, out, n_win)
            if (retry_ctr && threadIdx.x == 0) atomicAdd(retry_ctr, 1ull);
            eeg_one_window<NB, RES, W, WT>(smem, windows, win, n_ch, n_t, thresh, L, out, dist, corr);
        RETRY_SCAN_END()
    }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static inline int align16(int x) { return (x + 15) & ~15; }

// ---- lists of the windows to redo (tda_ctx::retry_buf) ----
tda_status retry_lists_reserve(tda_ctx* ctx, int n_win)
{
    if (n_win <= ctx->retry_cap) return TDA_OK;
    const int cap = (n_win + 1023) & ~1023;
    for (int i = 0; i < TDA_RETRY_SLOTS; ++i) {
        int* q = nullptr;
        TDA_HIP(ctx, hipMalloc((void**)&q, ((size_t)cap + 4) * sizeof(int)));
        TDA_HIP(ctx, hipMemset(q, 0, 4 * sizeof(int)));                        // (every pass leaves its list empty: see RETRY_SCAN_END)
        if (ctx->retry_buf[i]) ctx->retired.push_back(ctx->retry_buf[i]);     // (a captured graph may still name it)
        ctx->retry_buf[i] = q;
    }
    ctx->retry_cap = cap;
    return TDA_OK;
}
// the list of the stream the call is enqueued on (see tda_ctx::retry_buf)
static tda_status retry_list_take(tda_ctx* ctx, int n_win, RipsOut& out, hipStream_t st)
{
    const tda_status rc = retry_lists_reserve(ctx, n_win);
    if (rc != TDA_OK) return rc;
    int i = 0;
    while (i < ctx->retry_streams && ctx->retry_stream[i] != st) ++i;
    if (i == ctx->retry_streams) {
        if (i < TDA_RETRY_SLOTS) ctx->retry_stream[ctx->retry_streams++] = st;
        else { i = TDA_RETRY_SLOTS - 1; ++ctx->retry_shared; }
    }
    out.retry_list = ctx->retry_buf[i];
    return TDA_OK;
}
// before every widening pass: which windows are flagged now
static tda_status retry_collect(tda_ctx* ctx, const RipsOut& out, int n_win, hipStream_t st)
{
    hipLaunchKernelGGL(retry_collect_kernel, dim3((n_win + 255) / 256), dim3(256), 0, st, out.status, n_win, out.retry_list);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

static RipsLayout make_layout(int n, int psi_bytes_per_edge, int aux_bytes, int NT)   // classes = 8 * psi_bytes_per_edge
{
    RipsLayout L;
    const int E = n * (n - 1) / 2;
    const int psi_bytes = E * psi_bytes_per_edge;
    // ranking phase:  [key32 4E] .. [members 2E]      sweep phase:  [psi][rank 2E][ord 2E]
    // rank is written while the bucket members are still read, ord afterwards: with 4-byte class vectors the
    // members take the place of ord, otherwise they follow the keys (rank / ord lie beyond 8E then).
    // aux (point cloud / sorted keys) and misc follow; the bucket cursors (NB u16) lie over the part of misc that
    // only the sweep uses.
    L.off_rank = align16(psi_bytes) + GUARD_BYTES;
    L.off_ord = align16(L.off_rank + 2 * E) + GUARD_BYTES;
    L.off_members = psi_bytes_per_edge < 8 ? L.off_ord : 4 * E;
    int after_rank = align16(L.off_ord + 2 * E) + GUARD_BYTES;
    L.off_aux = after_rank;
    L.off_misc = align16(L.off_aux + aux_bytes) + GUARD_BYTES;
    int misc_bytes = MISC_BYTES(8 * psi_bytes_per_edge);
    const int nb = (NT == 256 && n <= 64) ? 2048 : (NT == 512 ? CLOUD_NB : 8192);
    if (misc_bytes < MISC_SORTCNT + 2 * nb) misc_bytes = MISC_SORTCNT + 2 * nb;
    L.total = align16(L.off_misc + misc_bytes) + GUARD_BYTES;
    L.off_psi = 0; L.psi_bytes = 0; L.off_cursor = L.off_misc + MISC_SORTCNT; L.off_rscr = L.off_misc;
    L.guard[0] = L.off_rank; L.guard[1] = L.off_ord; L.guard[2] = L.off_aux; L.guard[3] = L.off_misc; L.guard[4] = L.total;
    L.n_guard = 5;
    return L;
}

static RipsLayout make_narrow_layout(int n, int dim)      // total == 0: the cloud does not fit the fixed layout (struct NL)
{
    static_assert(MISC_BYTES(32) == 8752 + 8 * 32, "NL::MISC assumes this");
    RipsLayout L;
    L.off_rank = NL::RANK; L.off_psi = NL::PSI; L.off_members = NL::MEMBERS; L.off_cursor = NL::CURSOR; L.off_rscr = NL::RSCR;
    L.off_aux = NL::AUX; L.off_misc = NL::MISC; L.psi_bytes = NL::REGION; L.off_ord = 0;
    L.total = (n <= NARROW_PMAX && dim <= NARROW_DIM) ? NL::TOTAL : 0;
    L.guard[0] = NL::PSI; L.guard[1] = NL::MISC; L.guard[2] = NL::AUX; L.guard[3] = NL::TOTAL; L.guard[4] = NL::TOTAL;
    L.n_guard = 4;
    return L;
}

// H1 rows -> ripser's order (descending birth; ties: descending death, then emission order): the finishing pass of
// features.hip, unless the caller runs that pass itself for several diagram sets at once (tda_set_h1_order)
static tda_status order_h1(tda_ctx* ctx, double* h1, int h1_cap, int* h1_cnt, int n_win, hipStream_t st)
{
    if (h1_cap < 2 || ctx->h1_order == TDA_ORDER_DEFERRED) return TDA_OK;
    tda_diagram_set one{h1, h1_cnt, h1_cap, 1, nullptr};
    return launch_diagram_finish(ctx, &one, 1, n_win, st);
}

template <int NVW, int W>
static tda_status launch_dm_t(tda_ctx* ctx, const double* dm, int n_win, int n, float thresh, int symmetrise,
                              RipsOut out, hipStream_t st, int retry_only = 0)
{
    const int NT = 256;
    const RipsLayout L = make_layout(n, W * 8, n * (n - 1) / 2 * 4, NT);
    auto kern = retry_only ? rips_dm_kernel<256, NVW, W, u64, true> : rips_dm_kernel<256, NVW, W, u64, false>;
    if (L.total > 48 * 1024)
        TDA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    // retry passes work off the list of flagged windows on a grid that fits the chip; the widest variant (240 VGPRs, > 80 KB LDS)
    // needs a nearly empty CU per workgroup, so it asks for few of them
    const int rgrid = W >= 8 ? 64 : 512;
    const int grid = retry_only ? (n_win < rgrid ? n_win : rgrid) : n_win;
    if (retry_only) { const tda_status rc = retry_collect(ctx, out, n_win, st); if (rc != TDA_OK) return rc; }
    {
        ProbeScope probe(ctx, retry_only ? -1 : TDA_PROBE_RIPS_DM, st);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), L.total, st, dm, n_win, n, thresh, symmetrise, L, out,
                           ctx->retry_ctr);
    }
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

static const int LDS_MAX = 160 * 1024;

size_t rips_total_scratch_bytes() { return (size_t)TOT_SLOTS * TOT_SLOT_WORDS * sizeof(u64); }

// layout of the last rung: the wide layout with 32-bit class words (the area of the class vectors holds the birth lengths,
// one float per possible class) and 2 KB more behind the class tables for the stored / alive bit maps
static RipsLayout total_layout(int n, int aux_bytes, int NT)
{
    RipsLayout L = make_layout(n, 4, aux_bytes, NT);
    int need = align16(L.off_misc + MISC_BRANK + 2048) + GUARD_BYTES;
    const int need_rank = align16(L.off_misc + MISC_SORTCNT + 2 * 8192) + GUARD_BYTES;      // its ranking always has 8,192 buckets
    if (need < need_rank) need = need_rank;
    if (L.total < need) { L.total = need; L.guard[4] = L.total; }
    return L;
}
static tda_status launch_dm_total(tda_ctx* ctx, const double* dm, int n_win, int n, float thresh, int symmetrise, RipsOut out,
                                  hipStream_t st)
{
    const RipsLayout L = total_layout(n, n * (n - 1) / 2 * 4, 256);
    auto kern = rips_dm_total_kernel<256>;
    if (L.total > 48 * 1024)
        TDA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    { const tda_status rc = retry_collect(ctx, out, n_win, st); if (rc != TDA_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3(TOT_SLOTS), dim3(256), L.total, st, dm, n_win, n, thresh, symmetrise, L, out,
                       ctx->total_scratch, ctx->retry_ctr);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}
#ifndef CLOUD_NT
#define CLOUD_NT 512           // workgroup size of the point-cloud flavour (one workgroup per CU: LDS)
#endif

tda_status launch_rips_dm(tda_ctx* ctx, const double* dm, int n_win, int n, double thresh, int symmetrise,
                          double* h0, int h0_cap, int* h0_cnt, double* h1, int h1_cap, int* h1_cnt, int* status,
                          hipStream_t st)
{
    if (n_win == 0) return TDA_OK;
    if (n < 1 || n > TDA_MAX_POINTS) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "n must be in [1,128]");
    if (h0_cap < n) TDA_FAIL(ctx, TDA_ERR_INVALID, "h0_cap must be >= n");
    RipsOut out{h0, h0_cap, h0_cnt, h1, h1_cap, h1_cnt, status};
    const float th = (float)thresh;
    tda_status rc = retry_list_take(ctx, n_win, out, st);
    if (rc != TDA_OK) return rc;
    int W = ctx->words_dm < 1 ? 1 : ctx->words_dm;      // (32-bit class words exist in the fused EEG kernel only)
    // largest class capacity that still fits the 160 KiB LDS
    while (W > 1 && make_layout(n, W * 8, n * (n - 1) / 2 * 4, 256).total > LDS_MAX) W >>= 1;
    const bool last_only = ctx->retry_policy == TDA_RETRY_LAST_RUNG;
    const bool do_first = ctx->retry_policy != TDA_RETRY_ONLY && !last_only;
    const bool do_ladder = ctx->retry_policy != TDA_RETRY_FIRST_PASS && !last_only;
    const bool one_step = ctx->retry_policy == TDA_RETRY_ONE_STEP;
    rc = TDA_OK;
    if (last_only) {
    } else if (n <= 64) {
        if (!do_first) {}
        else if (W == 1) rc = launch_dm_t<1, 1>(ctx, dm, n_win, n, th, symmetrise, out, st);
        else if (W == 2) rc = launch_dm_t<1, 2>(ctx, dm, n_win, n, th, symmetrise, out, st);
        else rc = launch_dm_t<1, 4>(ctx, dm, n_win, n, th, symmetrise, out, st);
        // widening retries: windows that ran out of class bits are redone with 2x, 4x the bits while
        // the table still fits LDS (n <= 47: 512 bits cover the theoretical maximum of 506 alive
        // classes).  A retry launch whose windows are all fine exits at once.
        for (int Wr = 2 * W; do_ladder && rc == TDA_OK && Wr <= 8; Wr *= 2) {
            if (one_step && Wr > 2 * W) break;
            if (make_layout(n, Wr * 8, n * (n - 1) / 2 * 4, 256).total > LDS_MAX) break;
            if (Wr == 2) rc = launch_dm_t<1, 2>(ctx, dm, n_win, n, th, symmetrise, out, st, 1);
            else if (Wr == 4) rc = launch_dm_t<1, 4>(ctx, dm, n_win, n, th, symmetrise, out, st, 1);
            else rc = launch_dm_t<1, 8>(ctx, dm, n_win, n, th, symmetrise, out, st, 1);
        }
    } else {
        if (!do_first) {}
        else if (W == 1) rc = launch_dm_t<2, 1>(ctx, dm, n_win, n, th, symmetrise, out, st);
        else rc = launch_dm_t<2, 2>(ctx, dm, n_win, n, th, symmetrise, out, st);
        if (do_ladder && rc == TDA_OK && W == 1 && make_layout(n, 16, n * (n - 1) / 2 * 4, 256).total <= LDS_MAX)
            rc = launch_dm_t<2, 2>(ctx, dm, n_win, n, th, symmetrise, out, st, 1);
    }
    // the last rung: whatever is still flagged (more than 512 / 128 classes alive at once) is redone with the class
    // vectors in HBM -- compute_eeg_persistence never refuses a matrix (scripts/utils.py:140)
    if (rc == TDA_OK && ((do_ladder && !one_step) || last_only)) rc = launch_dm_total(ctx, dm, n_win, n, th, symmetrise, out, st);
    if (rc != TDA_OK) return rc;
    return order_h1(ctx, h1, h1_cap, h1_cnt, n_win, st);
}

template <int NB, bool RES, int W, bool RETRY, typename WT = u64>
static tda_status launch_eeg_t(tda_ctx* ctx, const WindowSource& win, int n_win, int n_ch, int n_t, float thresh, RipsOut out,
                               double* dist, double* corr, hipStream_t st)
{
    RipsLayout L = make_layout(n_ch, W * (int)sizeof(WT), n_ch * (n_ch - 1) / 2 * 4, 256);
    if ((size_t)L.total < CdLayout<NB>::BYTES) L.total = (int)((CdLayout<NB>::BYTES + 15) & ~(size_t)15);
    if (L.total > LDS_MAX) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "window too large for LDS");
    auto kern = eeg_window_kernel<NB, RES, W, RETRY, WT>;
    if (L.total > 48 * 1024)
        TDA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    const int rgrid = W >= 8 ? 64 : 512;
    const int grid = RETRY ? (n_win < rgrid ? n_win : rgrid) : n_win;
    if (RETRY) { const tda_status rc = retry_collect(ctx, out, n_win, st); if (rc != TDA_OK) return rc; }
    {
        ProbeScope probe(ctx, RETRY ? -1 : TDA_PROBE_RIPS_DM, st);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), L.total, st, win, n_win, n_ch, n_t, thresh, L, out, dist, corr,
                           ctx->retry_ctr);
    }
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

template <bool RES>
static tda_status launch_eeg_ladder(tda_ctx* ctx, const WindowSource& win, int n_win, int n_ch, int n_t, float th, RipsOut out,
                                    double* dist, double* corr, hipStream_t st)
{
    if (ctx->retry_policy == TDA_RETRY_LAST_RUNG) return TDA_OK;     // (512 bits cover every complex on 48 points: no such rung here)
    const bool do_first = ctx->retry_policy != TDA_RETRY_ONLY, do_ladder = ctx->retry_policy != TDA_RETRY_FIRST_PASS;
    const bool one_step = ctx->retry_policy == TDA_RETRY_ONE_STEP;
    tda_status rc = TDA_OK;
    // ladder of the fused kernel: 32 (words_dm = 0) -> 64 -> 128 -> 512 class bits; every wider pass recomputes the
    // flagged windows from the samples (the matrix was never stored); 512 covers the theoretical maximum of 506
    // classes alive at once for 47 points
    const int first = ctx->words_dm >= 2 ? 2 : ctx->words_dm;       // 0, 1 or 2
    if (do_first) {
        if (first == 0) rc = launch_eeg_t<3, RES, 1, false, u32>(ctx, win, n_win, n_ch, n_t, th, out, dist, corr, st);
        else if (first == 1) rc = launch_eeg_t<3, RES, 1, false>(ctx, win, n_win, n_ch, n_t, th, out, dist, corr, st);
        else rc = launch_eeg_t<3, RES, 2, false>(ctx, win, n_win, n_ch, n_t, th, out, dist, corr, st);
    }
    int rung = 0;
    if (do_ladder && rc == TDA_OK && first == 0 && !(one_step && rung++ >= 1))
        rc = launch_eeg_t<3, RES, 1, true>(ctx, win, n_win, n_ch, n_t, th, out, dist, corr, st);
    if (do_ladder && rc == TDA_OK && first <= 1 && !(one_step && rung++ >= 1))
        rc = launch_eeg_t<3, RES, 2, true>(ctx, win, n_win, n_ch, n_t, th, out, dist, corr, st);
    if (do_ladder && rc == TDA_OK && !(one_step && rung++ >= 1))
        rc = launch_eeg_t<3, RES, 8, true>(ctx, win, n_win, n_ch, n_t, th, out, dist, corr, st);
    return rc;
}

static tda_status launch_eeg_source(tda_ctx* ctx, const WindowSource& src, int n_win, int n_ch, int n_t, double thresh,
                                    double* dist, double* corr, double* h0, int h0_cap, int* h0_cnt, double* h1, int h1_cap,
                                    int* h1_cnt, int* status, hipStream_t st)
{
    if (n_win == 0) return TDA_OK;
    if (n_ch < 33 || n_ch > 48) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "the fused EEG kernel takes 33..48 channels (the reference has 47); use tda_corr_dist_batch + tda_rips_dm_batch otherwise");
    if (n_t < 2 || n_t > CD_RES_CHUNKS * CD_TCH) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "the fused EEG kernel takes windows of 2..256 samples (the reference has 250)");
    if (h0_cap < n_ch) TDA_FAIL(ctx, TDA_ERR_INVALID, "h0_cap must be >= n_ch");
    RipsOut out{h0, h0_cap, h0_cnt, h1, h1_cap, h1_cnt, status};
    { const tda_status rc0 = retry_list_take(ctx, n_win, out, st); if (rc0 != TDA_OK) return rc0; }
    // The window is fetched twice (means, then centred products): 128 VGPRs, four workgroups per CU.  The second
    // fetch misses the 4 MB L2 of the XCD (128 windows in flight there) and shows in FETCH_SIZE (181 KB per window
    // against 95 KB algorithmic), but the windows in flight on the whole chip are 96 MB, inside the 256 MiB Infinity
    // Cache, which that counter does not tell from HBM.  Keeping the window in registers between the passes instead
    // (one fetch, 148 VGPRs, three workgroups per CU) measured 11 % slower on the EEG-only pass and 2 % end to end;
    // it stays available: TDA_EEG_RESIDENT=1
    static const bool stream_twice = getenv("TDA_EEG_RESIDENT") == nullptr;
    const tda_status rc = stream_twice ? launch_eeg_ladder<false>(ctx, src, n_win, n_ch, n_t, (float)thresh, out, dist, corr, st)
                                       : launch_eeg_ladder<true>(ctx, src, n_win, n_ch, n_t, (float)thresh, out, dist, corr, st);
    if (rc != TDA_OK) return rc;
    return order_h1(ctx, h1, h1_cap, h1_cnt, n_win, st);
}

tda_status launch_eeg_windows(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t, double thresh, double* dist,
                              double* corr, double* h0, int h0_cap, int* h0_cnt, double* h1, int h1_cap, int* h1_cnt,
                              int* status, hipStream_t st)
{
    const WindowSource src{win, (long long)n_ch * n_t, 0, n_t, 0x7fffffff, nullptr};
    return launch_eeg_source(ctx, src, n_win, n_ch, n_t, thresh, dist, corr, h0, h0_cap, h0_cnt, h1, h1_cap, h1_cnt, status, st);
}

// windows read in place from band-passed recordings of equal length (n_rec, n_ch, n_samples); sel: optional window list
tda_status launch_eeg_sliding(tda_ctx* ctx, const double* sig, int n_rec, int n_ch, int n_samples, int win_len, int step,
                              const int* sel, int n_sel, double thresh, double* dist, double* corr, double* h0, int h0_cap,
                              int* h0_cnt, double* h1, int h1_cap, int* h1_cnt, int* status, int* n_win_out, hipStream_t st)
{
    if (win_len < 2 || step < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "win_len must be >= 2 and step >= 1");
    const int per_rec = n_samples >= win_len ? (n_samples - win_len) / step + 1 : 0;      // nb1:341
    if (n_win_out) *n_win_out = per_rec;
    if (n_rec <= 0 || per_rec == 0) return TDA_OK;
    if ((long long)n_rec * per_rec > 0x7fffffffLL) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "too many windows in one call");
    const int n_out = sel ? n_sel : n_rec * per_rec;
    const WindowSource src{sig, (long long)step, (long long)n_ch * n_samples, n_samples, per_rec, sel};
    return launch_eeg_source(ctx, src, n_out, n_ch, win_len, thresh, dist, corr, h0, h0_cap, h0_cnt, h1, h1_cap, h1_cnt, status, st);
}

template <int W, typename WT, bool NARROW = false>
static tda_status launch_cloud_t(tda_ctx* ctx, const double* src, const int* aux, int n_win, int n_t_or_pcap,
                                 int dim, int subsample, int mode, int normalise, float thresh, int p_max,
                                 int* n_points, RipsOut out, hipStream_t st, int retry_only)
{
    const int NT = NARROW ? NARROW_NT : CLOUD_NT;
    const RipsLayout L = NARROW ? make_narrow_layout(p_max, dim)
                                : make_layout(p_max, W * (int)sizeof(WT), p_max * dim * 8, NT);
    if (L.total > LDS_MAX || L.total == 0) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "point cloud too large for LDS");
    auto kern = retry_only ? rips_cloud_kernel<CLOUD_NT, W, WT, true, false>
                           : rips_cloud_kernel<(NARROW ? NARROW_NT : CLOUD_NT), W, WT, false, NARROW>;
    if (L.total > 48 * 1024)
        TDA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    const int rgrid = W >= 2 ? 256 : 512;              // (what fits the chip at once: two / one workgroup per CU)
    const int grid = retry_only ? (n_win < rgrid ? n_win : rgrid) : n_win;
    if (retry_only) { const tda_status rc = retry_collect(ctx, out, n_win, st); if (rc != TDA_OK) return rc; }
    {
        ProbeScope probe(ctx, retry_only ? -1 : TDA_PROBE_RIPS_CLOUD, st);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), L.total, st, src, aux, n_win, n_t_or_pcap, dim, subsample, mode,
                           normalise, thresh, L, p_max, n_points, out, probe.on ? probe.span : nullptr,
                           ctx->retry_ctr);
    }
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

tda_status launch_rips_cloud(tda_ctx* ctx, const double* src, const int* aux, int n_win, int n_t_or_pcap, int dim,
                             int subsample, int mode, int normalise, double thresh, double* h0, int h0_cap,
                             int* h0_cnt, double* h1, int h1_cap, int* h1_cnt, int* n_points, int* status,
                             hipStream_t st)
{
    if (n_win == 0) return TDA_OK;
    if (dim < 1 || dim > TDA_MAX_DIM) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "dim must be in [1,4]");
    if (subsample < 1) subsample = 1;
    int p_max;
    if (mode == 0) {
        // tau >= 1 in the reference (utils.py:103-104); P is largest at tau = 1
        const int nn = n_t_or_pcap - (dim - 1);
        p_max = nn > 0 ? (nn + subsample - 1) / subsample : 0;
    } else {
        p_max = n_t_or_pcap;
    }
    if (p_max > TDA_MAX_POINTS) p_max = TDA_MAX_POINTS;   // larger clouds are flagged per window
#ifdef TDA_EXPERIMENT
    if (getenv("TDA_EXP_PMAX")) p_max = atoi(getenv("TDA_EXP_PMAX"));   // co-residency probe: smaller LDS layout
#endif
    if (p_max < 3) p_max = 3;
    if (h0_cap < p_max) TDA_FAIL(ctx, TDA_ERR_INVALID, "h0_cap must be >= max points per cloud");
    RipsOut out{h0, h0_cap, h0_cnt, h1, h1_cap, h1_cnt, status};
    const float th = (float)thresh;
    tda_status rc = retry_list_take(ctx, n_win, out, st);
    if (rc != TDA_OK) return rc;
    // class capacity ladder: 32 bits (two workgroups per CU for 124-point clouds), then 64, then 128
    // while the table fits LDS; each wider pass only redoes the windows the previous one flagged
    const bool fits128 = make_layout(p_max, 16, p_max * dim * 8, CLOUD_NT).total <= LDS_MAX;
    const bool last_only = ctx->retry_policy == TDA_RETRY_LAST_RUNG;
    const bool do_first = ctx->retry_policy != TDA_RETRY_ONLY && !last_only;
    const bool do_ladder = ctx->retry_policy != TDA_RETRY_FIRST_PASS && !last_only;
    int first = 1;
    if (ctx->words_cloud == 1) {
        // first pass: 32 class bits, the narrow layout (three workgroups per CU) whenever the cloud size allows it
        static const bool no_narrow = getenv("TDA_CLOUD_WIDE_FIRST") != nullptr;
        const bool narrow = !no_narrow && make_narrow_layout(p_max, dim).total != 0;
        if (do_first && narrow)
            rc = launch_cloud_t<1, u32, true>(ctx, src, aux, n_win, n_t_or_pcap, dim, subsample, mode, normalise, th, p_max,
                                              n_points, out, st, 0);
        else if (do_first)
            rc = launch_cloud_t<1, u32>(ctx, src, aux, n_win, n_t_or_pcap, dim, subsample, mode, normalise, th, p_max,
                                        n_points, out, st, 0);
        first = 0;
    }
    if (rc == TDA_OK && (first ? do_first : do_ladder))
        rc = launch_cloud_t<1, u64>(ctx, src, aux, n_win, n_t_or_pcap, dim, subsample, mode, normalise, th, p_max,
                                    n_points, out, st, first ? 0 : 1);
    if (do_ladder && rc == TDA_OK && fits128 && !(ctx->retry_policy == TDA_RETRY_ONE_STEP && !first))
        rc = launch_cloud_t<2, u64>(ctx, src, aux, n_win, n_t_or_pcap, dim, subsample, mode, normalise, th, p_max,
                                    n_points, out, st, 1);
    // the last rung: whatever is still flagged (more than 64 / 128 classes alive at once) is redone with the class vectors
    // in HBM -- compute_audio_persistence never refuses a cloud (scripts/utils.py:131)
    if (rc == TDA_OK && ((do_ladder && ctx->retry_policy != TDA_RETRY_ONE_STEP) || last_only)) {
        const RipsLayout L = total_layout(p_max, p_max * dim * 8, CLOUD_NT);
        auto kern = rips_cloud_total_kernel<CLOUD_NT>;
        if (L.total > 48 * 1024)
            TDA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
        rc = retry_collect(ctx, out, n_win, st);
        if (rc != TDA_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(TOT_SLOTS), dim3(CLOUD_NT), L.total, st, src, aux, n_win, n_t_or_pcap, dim, subsample, mode,
                           normalise, th, L, p_max, n_points, out, ctx->total_scratch, ctx->retry_ctr);
        TDA_HIP(ctx, hipGetLastError());
    }
    if (rc != TDA_OK) return rc;
    return order_h1(ctx, h1, h1_cap, h1_cnt, n_win, st);
}

// rips.hip -- batched Vietoris-Rips H0/H1 persistence for gfx950 (MI355X), wave64.
//
// Replaces ripser(dm, maxdim=1, thresh, distance_matrix=True) at scripts/utils.py:140 /
// scripts/tda_eeg_classification_v2.py:170-175 and ripser(pc_norm, maxdim=1, thresh) at
// scripts/utils.py:131 (incl. the Takens embedding utils.py:107-116 and the min-max
// normalisation utils.py:127-130 in front of it).
//
// One window per 64-thread workgroup (= one wavefront, so every step is wave-synchronous
// and needs no s_barrier).  Algorithm (NOT ripser's; designed for a single wave):
//
//  1. all n(n-1)/2 float32 edge lengths are packed as (sortable key << 16 | a << 8 | b)
//     and bitonic-sorted in LDS.
//  2. one sweep over the edges in filtration order keeps, in registers,
//       * adj[v]   : adjacency bit rows of the graph so far (lane v <-> vertex v),
//       * comp[v]  : connected-component label (H0 by label propagation = Kruskal),
//       * for every currently alive H1 class one bit; brank/bkey[bit] live in lane `bit`;
//     and in LDS psi[edge] = the class of the cycle "edge + tree path" as a bit vector
//     over alive classes.  For edge e=(a,b) with common-neighbour mask M = adj[a]&adj[b]:
//       M == 0, different components : H0 death at |e|            (negative edge)
//       M == 0, same component       : a new H1 class is born     (positive edge)
//       M != 0                       : e is killed at once by the triangle (a,b,v*),
//             v* = lowest vertex of M:  psi[e] = psi[a,v*] ^ psi[b,v*]; every other
//             triangle (a,b,v), v in M, has boundary class psi[e]^psi[a,v]^psi[b,v]; if
//             that is non-zero the YOUNGEST class in it dies at |e| and is substituted
//             out of the whole psi table (elder rule).  Such kills happen exactly once
//             per off-diagonal H1 point, so the table pass is rare.
//     The multiset of (birth,death) pairs equals that of any persistence algorithm on the
//     same filtration; tie order inside equal diameters does not change it.
//  3. rows are written as float64 (float32-exact) pairs: H0 ascending death then the
//     essential rows; H1 rows are ordered by a second tiny kernel (descending birth).
//
// LDS per workgroup: 23.8 KB (n = 47, 128 classes) .. 80.2 KB (n = 124 point cloud), see
// rips_lds_bytes().  No MFMA: this is irregular integer work.
#include "common.h"

#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

// ---- optional phase profiling (make PROFILE=1): cycle sums per phase over all windows ----
#ifdef TDA_PROFILE
__device__ unsigned long long g_prof[16];
#define PROF_BEGIN() unsigned long long prof_t0 = clock64()
#define PROF_MARK(i)                                                       \
    do {                                                                   \
        unsigned long long prof_t1 = clock64();                            \
        if (lane_id() == 0) atomicAdd(&g_prof[i], prof_t1 - prof_t0);      \
        prof_t0 = prof_t1;                                                 \
    } while (0)
#define PROF_COUNT(i, v) do { if (lane_id() == 0) atomicAdd(&g_prof[i], (unsigned long long)(v)); } while (0)
extern "C" __attribute__((visibility("default"))) int tda_profile_read(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#else
#define PROF_BEGIN() do {} while (0)
#define PROF_MARK(i) do {} while (0)
#define PROF_COUNT(i, v) do {} while (0)
#endif

template <int W>
struct Psi {
    u64 w[W];
};

template <int W>
__device__ __forceinline__ Psi<W> pxor(Psi<W> a, Psi<W> b)
{
    Psi<W> r;
#pragma unroll
    for (int i = 0; i < W; ++i) r.w[i] = a.w[i] ^ b.w[i];
    return r;
}
template <int W>
__device__ __forceinline__ bool pnz(Psi<W> a)
{
    u64 o = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) o |= a.w[i];
    return o != 0;
}
template <int W>
__device__ __forceinline__ Psi<W> pzero()
{
    Psi<W> r;
#pragma unroll
    for (int i = 0; i < W; ++i) r.w[i] = 0;
    return r;
}
template <int W>
__device__ __forceinline__ Psi<W> prl(Psi<W> a, int lane)
{
    Psi<W> r;
#pragma unroll
    for (int i = 0; i < W; ++i) r.w[i] = rl64(a.w[i], lane);
    return r;
}

// "write lane": this clang has no writelane builtin; a compare+select is 3 VALU ops
__device__ __forceinline__ u64 wl64(u64 val, int lane, u64 old)
{
    return (lane_id() == lane) ? val : old;
}

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ int tri2(int v) { return (v * (v - 1)) >> 1; }

// in-LDS bitonic sort of npad (power of two) u64 keys by one wave
__device__ void bitonic_sort_lds(u64* S, int npad)
{
    const int lane = lane_id();
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (npad >> 1); t += 64) {
                // t-th compare-exchange of this stage: i has bit j clear
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int l = i | j;
                u64 x = S[i], y = S[l];
                bool up = (i & k) == 0;
                bool sw = up ? (x > y) : (x < y);
                if (sw) { S[i] = y; S[l] = x; }
            }
            WAVE_SYNC();
        }
    }
}

struct RipsOut {
    double* h0; int h0_cap; int* h0_cnt;
    double* h1; int h1_cap; int* h1_cnt;
    int* status;
};

// ---------------------------------------------------------------------------------
// The sweep.  KEYFN(r, a, b) returns the float32 length of sorted edge r = (a,b).
// ---------------------------------------------------------------------------------
template <int NVW, int W, class KEYFN>
__device__ void rips_sweep(int n, int E, int Ev, const u16* ord, Psi<W>* psi, KEYFN keyfn,
                           double* h0, int h0_cap, double* h1, int h1_cap,
                           int& out_k0, int& out_k1, int& out_status)
{
    const int lane = lane_id();
    // zero the class table
    for (int e = lane; e < E; e += 64) psi[e] = pzero<W>();
    WAVE_SYNC();

    u64 adj[NVW][NVW];   // adj[set][word]: rows lane+64*set, bits of vertices 64*word..
    int comp[NVW];
    int tv[NVW];
#pragma unroll
    for (int s = 0; s < NVW; ++s) {
#pragma unroll
        for (int w = 0; w < NVW; ++w) adj[s][w] = 0;
        comp[s] = lane + 64 * s;
        tv[s] = tri2(lane + 64 * s);
    }
    u64 alive[W];
    int brank[W];
    float bkey[W];
#pragma unroll
    for (int c = 0; c < W; ++c) { alive[c] = 0; brank[c] = -1; bkey[c] = 0.f; }

    int k0 = 0, k1 = 0, merges = 0, status = 0;

    for (int r0 = 0; r0 < Ev && !(status & TDA_WIN_CLASS_OVERFLOW); r0 += 64) {
        const int cnt = (Ev - r0) < 64 ? (Ev - r0) : 64;
        const u32 ochunk = (r0 + lane < Ev) ? (u32)ord[r0 + lane] : 0u;
        for (int q = 0; q < cnt; ++q) {
            const int r = r0 + q;
            const u32 pk = rl32(ochunk, q);
            const int a = (int)(pk >> 8), b = (int)(pk & 255u);   // a > b
            const int ta = tri2(a), tb = tri2(b);
            const int tab = ta + b;
            // adjacency rows of a and b
            u64 ra[NVW], rb[NVW];
            if (NVW == 1) {
                ra[0] = rl64(adj[0][0], a);
                rb[0] = rl64(adj[0][0], b);
            } else {
#pragma unroll
                for (int w = 0; w < NVW; ++w) {
                    ra[w] = (a < 64) ? rl64(adj[0][w], a) : rl64(adj[NVW - 1][w], a - 64);
                    rb[w] = (b < 64) ? rl64(adj[0][w], b) : rl64(adj[NVW - 1][w], b - 64);
                }
            }
            u64 mask[NVW];
            u64 many = 0;
#pragma unroll
            for (int w = 0; w < NVW; ++w) { mask[w] = ra[w] & rb[w]; many |= mask[w]; }

            if (many == 0) {
                const float key = keyfn(r, a, b);
                int ca, cb;
                if (NVW == 1) { ca = (int)rl32((u32)comp[0], a); cb = (int)rl32((u32)comp[0], b); }
                else {
                    ca = (a < 64) ? (int)rl32((u32)comp[0], a) : (int)rl32((u32)comp[NVW - 1], a - 64);
                    cb = (b < 64) ? (int)rl32((u32)comp[0], b) : (int)rl32((u32)comp[NVW - 1], b - 64);
                }
                if (ca != cb) {
                    // negative edge: two components merge, H0 class dies at |e|
#pragma unroll
                    for (int s = 0; s < NVW; ++s) comp[s] = (comp[s] == cb) ? ca : comp[s];
                    ++merges;
                    if (key != 0.0f) {
                        if (k0 < h0_cap && lane == 0) { h0[2 * k0] = 0.0; h0[2 * k0 + 1] = (double)key; }
                        ++k0;
                    }
                } else {
                    // positive edge with no apex yet: a new H1 class
                    int cw = -1, bit = 0;
#pragma unroll
                    for (int c = W - 1; c >= 0; --c) {
                        u64 fr = ~alive[c];
                        if (fr) { cw = c; bit = __builtin_ctzll(fr); }
                    }
                    if (cw < 0) { status |= TDA_WIN_CLASS_OVERFLOW; break; }
                    Psi<W> nv = pzero<W>();
#pragma unroll
                    for (int c = 0; c < W; ++c)
                        if (c == cw) {
                            alive[c] |= (1ull << bit);
                            nv.w[c] = (1ull << bit);
                            if (lane == bit) { brank[c] = r; bkey[c] = key; }
                        }
                    if (lane == 0) psi[tab] = nv;
                }
            } else {
                // apparent edge: killed by triangle (a,b,v*), then test the other triangles
                Psi<W> x[NVW];
                bool act[NVW];
#pragma unroll
                for (int w = 0; w < NVW; ++w) {
                    const int v = lane + 64 * w;
                    act[w] = (mask[w] >> lane) & 1ull;
                    x[w] = pzero<W>();
                    if (act[w]) {
                        const int ia = (v < a) ? ta + v : tv[w] + a;
                        const int ib = (v < b) ? tb + v : tv[w] + b;
                        x[w] = pxor(psi[ia], psi[ib]);
                    }
                }
                Psi<W> base;
                if (NVW == 1 || mask[0]) base = prl(x[0], __builtin_ctzll(mask[0]));
                else base = prl(x[NVW - 1], __builtin_ctzll(mask[NVW - 1]));
                if (lane == 0) psi[tab] = base;
                bool nz = false;
#pragma unroll
                for (int w = 0; w < NVW; ++w) nz |= act[w] && pnz(pxor(x[w], base));
                if (__ballot(nz)) {
                    // ---- rare path: some triangle's boundary is a non-trivial class ----
#ifdef TDA_PROFILE
                    unsigned long long kt0 = clock64();
#endif
                    const float key = keyfn(r, a, b);
                    for (int guard = 0; guard < 64 * W + 2; ++guard) {
                        WAVE_SYNC();
                        const Psi<W> pe = psi[tab];
                        Psi<W> wv = pzero<W>();
                        bool found = false;
#pragma unroll
                        for (int w = 0; w < NVW; ++w) {
                            const int v = lane + 64 * w;
                            Psi<W> y = pzero<W>();
                            if (act[w]) {
                                const int ia = (v < a) ? ta + v : tv[w] + a;
                                const int ib = (v < b) ? tb + v : tv[w] + b;
                                y = pxor(pxor(psi[ia], psi[ib]), pe);
                            }
                            const u64 bal = __ballot(act[w] && pnz(y));
                            if (!found && bal) { found = true; wv = prl(y, __builtin_ctzll(bal)); }
                        }
                        if (!found) break;
                        // youngest class of wv (largest birth rank)
                        int cand = -1;
#pragma unroll
                        for (int c = 0; c < W; ++c)
                            if ((wv.w[c] >> lane) & 1ull) cand = brank[c] > cand ? brank[c] : cand;
                        const int ymax = wave_max_i32(cand);
                        int ycw = 0, ybit = 0;
                        float ybirth = 0.f;
#pragma unroll
                        for (int c = 0; c < W; ++c) {
                            const u64 bal = __ballot(((wv.w[c] >> lane) & 1ull) && brank[c] == ymax);
                            if (bal) { ycw = c; ybit = __builtin_ctzll(bal); ybirth = __uint_as_float(rl32(__float_as_uint(bkey[c]), ybit)); }
                        }
                        if (key > ybirth) {
                            if (k1 < h1_cap && lane == 0) { h1[2 * k1] = (double)ybirth; h1[2 * k1 + 1] = (double)key; }
                            ++k1;
                        }
                        // substitute the dead class out of the table
                        for (int e = lane; e < E; e += 64) {
                            Psi<W> p = psi[e];
                            u64 sel = 0;
#pragma unroll
                            for (int c = 0; c < W; ++c)
                                if (c == ycw) sel = (p.w[c] >> ybit) & 1ull;
                            if (sel) psi[e] = pxor(p, wv);
                        }
#pragma unroll
                        for (int c = 0; c < W; ++c)
                            if (c == ycw) {
                                alive[c] &= ~(1ull << ybit);
                                if (lane == ybit) brank[c] = -1;
                            }
                        PROF_COUNT(10, 1);
                    }
#ifdef TDA_PROFILE
                    PROF_COUNT(4, clock64() - kt0);
                    PROF_COUNT(11, 1);
#endif
                }
            }
            // insert the edge into the graph
            if (NVW == 1) {
                adj[0][0] = wl64(ra[0] | (1ull << b), a, adj[0][0]);
                adj[0][0] = wl64(rb[0] | (1ull << a), b, adj[0][0]);
            } else {
#pragma unroll
                for (int w = 0; w < NVW; ++w) {
                    const u64 na = ra[w] | (((b >> 6) == w) ? (1ull << (b & 63)) : 0ull);
                    const u64 nb = rb[w] | (((a >> 6) == w) ? (1ull << (a & 63)) : 0ull);
                    if (a < 64) adj[0][w] = wl64(na, a, adj[0][w]);
                    else adj[NVW - 1][w] = wl64(na, a - 64, adj[NVW - 1][w]);
                    if (b < 64) adj[0][w] = wl64(nb, b, adj[0][w]);
                    else adj[NVW - 1][w] = wl64(nb, b - 64, adj[NVW - 1][w]);
                }
            }
            WAVE_SYNC();
        }
    }
    // essential classes
    const int ncomp = n - merges;
    for (int i = 0; i < ncomp; ++i) {
        if (k0 < h0_cap && lane == 0) { h0[2 * k0] = 0.0; h0[2 * k0 + 1] = (double)INFINITY; }
        ++k0;
    }
#pragma unroll
    for (int c = 0; c < W; ++c) {
        u64 al = alive[c];
        while (al) {
            const int bit = __builtin_ctzll(al);
            al &= al - 1;
            const float bk = __uint_as_float(rl32(__float_as_uint(bkey[c]), bit));
            if (k1 < h1_cap && lane == 0) { h1[2 * k1] = (double)bk; h1[2 * k1 + 1] = (double)INFINITY; }
            ++k1;
        }
    }
    if (k1 > h1_cap) status |= TDA_WIN_H1_TRUNCATED;
    out_k0 = k0; out_k1 = k1; out_status = status;
}

// ---------------------------------------------------------------------------------
// distance-matrix flavour (EEG): LDS = [S | psi] [ord u16] [skey u32]
// ---------------------------------------------------------------------------------
struct KeyFromLds {
    const u32* skey;
    __device__ __forceinline__ float operator()(int r, int, int) const
    {
        return sortable_f32((u32)uni((int)skey[r]));
    }
};

template <int NVW, int W>
__global__ void __launch_bounds__(64)
rips_dm_kernel(const double* __restrict__ dm, int n_win, int n, float thresh, int symmetrise,
               int off_ord, int off_key, RipsOut out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int win = blockIdx.x;
    if (win >= n_win) return;
    const int lane = lane_id();
    const int E = tri2(n);
    int npad = 64;
    while (npad < E) npad <<= 1;
    u64* S = reinterpret_cast<u64*>(smem);
    u16* ord = reinterpret_cast<u16*>(smem + off_ord);
    u32* skey = reinterpret_cast<u32*>(smem + off_key);
    Psi<W>* psi = reinterpret_cast<Psi<W>*>(smem);

    const double* D = dm + (size_t)win * n * n;
    const u32 tkey = f32_sortable(thresh);
    int Ev = 0;
    // 1. keys: utils.py:137-139 then ripser's float32 cast
    for (int a = 1; a < n; ++a) {
        const int base = tri2(a);
        for (int b = lane; b < a; b += 64) {
            double v;
            if (symmetrise) {
                v = (D[(size_t)a * n + b] + D[(size_t)b * n + a]) / 2.0;
                if (v < 0.0) v = 0.0;
            } else {
                v = D[(size_t)b * n + a];
            }
            const u32 sk = f32_sortable((float)v);
            S[base + b] = ((u64)sk << 16) | (u64)((a << 8) | b);
            Ev += (sk <= tkey) ? 1 : 0;
        }
    }
    for (int e = E + lane; e < npad; e += 64) S[e] = ~0ull;
    // wave-sum of Ev
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) Ev += __shfl_xor(Ev, off, 64);
    Ev = uni(Ev);
    WAVE_SYNC();
    bitonic_sort_lds(S, npad);
    // 2. unpack (ord/skey live outside the S region)
    for (int e = lane; e < E; e += 64) {
        const u64 c = S[e];
        ord[e] = (u16)(c & 0xffffu);
        skey[e] = (u32)(c >> 16);
    }
    WAVE_SYNC();
    int k0, k1, st;
    KeyFromLds kf{skey};
    rips_sweep<NVW, W>(n, E, Ev, ord, psi, kf,
                       out.h0 + (size_t)win * out.h0_cap * 2, out.h0_cap,
                       out.h1 + (size_t)win * out.h1_cap * 2, out.h1_cap, k0, k1, st);
    if (lane == 0) { out.h0_cnt[win] = k0; out.h1_cnt[win] = k1; out.status[win] = st; }
}

// ---------------------------------------------------------------------------------
// point-cloud flavour (audio): LDS = [S -> ord in place | psi] [pts f64]
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

template <int W>
__global__ void __launch_bounds__(64)
rips_cloud_kernel(const double* __restrict__ src, const int* __restrict__ tau_or_npts, int n_win,
                  int n_t_or_pcap, int dim, int subsample, int mode, int normalise, float thresh,
                  int off_psi, int off_pts, int p_max, int* __restrict__ n_points, RipsOut out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int win = blockIdx.x;
    if (win >= n_win) return;
    const int lane = lane_id();
    u64* S = reinterpret_cast<u64*>(smem);
    u16* ord = reinterpret_cast<u16*>(smem);
    Psi<W>* psi = reinterpret_cast<Psi<W>*>(smem + off_psi);
    double* pts = reinterpret_cast<double*>(smem + off_pts);

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
    if (n_points && lane == 0) n_points[win] = P;
    if (P > p_max) {
        if (lane == 0) { out.h0_cnt[win] = 0; out.h1_cnt[win] = 0; out.status[win] = TDA_WIN_TOO_LARGE; }
        return;
    }
    if (P < 3) {
        // utils.py:125-126: [[0,0]], [[0,0]]
        if (lane == 0) {
            if (out.h0_cap > 0) { h0[0] = 0.0; h0[1] = 0.0; }
            if (out.h1_cap > 0) { h1[0] = 0.0; h1[1] = 0.0; }
            out.h0_cnt[win] = 1; out.h1_cnt[win] = 1;
            out.status[win] = TDA_WIN_DEGENERATE;
        }
        return;
    }
    PROF_BEGIN();
    // 1. cloud -> LDS, per-column min-max to [0,1] (range 0 -> 1)
    for (int k = 0; k < dim; ++k) {
        double mn = INFINITY, mx = -INFINITY;
        for (int i = lane; i < P; i += 64) {
            const double v = (mode == 0) ? base[i * subsample + k * tau] : base[i * dim + k];
            pts[i * dim + k] = v;
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
        if (normalise) {
            mn = wave_min_f64(mn);
            mx = wave_max_f64(mx);
            double rg = mx - mn;
            if (rg == 0.0) rg = 1.0;
            for (int i = lane; i < P; i += 64) pts[i * dim + k] = (pts[i * dim + k] - mn) / rg;
        }
    }
    WAVE_SYNC();
    // 2. keys
    const int E = tri2(P);
    int npad = 64;
    while (npad < E) npad <<= 1;
    const u32 tkey = f32_sortable(thresh);
    KeyFromPts kf{pts, dim};
    int Ev = 0;
    for (int a = 1; a < P; ++a) {
        const int tb = tri2(a);
        for (int b = lane; b < a; b += 64) {
            const u32 sk = f32_sortable(kf(0, a, b));
            S[tb + b] = ((u64)sk << 16) | (u64)((a << 8) | b);
            Ev += (sk <= tkey) ? 1 : 0;
        }
    }
    for (int e = E + lane; e < npad; e += 64) S[e] = ~0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) Ev += __shfl_xor(Ev, off, 64);
    Ev = uni(Ev);
    WAVE_SYNC();
    PROF_MARK(0);
    bitonic_sort_lds(S, npad);
    PROF_MARK(1);
    // 3. compact S -> ord in place, chunk by chunk (chunk r0 writes bytes [2 r0, 2 r0 + 128),
    //    all of which belong to entries < r0 + 16 <= already loaded)
    for (int r0 = 0; r0 < E; r0 += 64) {
        const int e = r0 + lane;
        const u64 c = (e < E) ? S[e] : 0ull;
        WAVE_SYNC();
        if (e < E) ord[e] = (u16)(c & 0xffffu);
        WAVE_SYNC();
    }
    int k0, k1, st;
    PROF_MARK(2);
    if (P <= 64)
        rips_sweep<1, W>(P, E, Ev, ord, psi, kf, h0, out.h0_cap, h1, out.h1_cap, k0, k1, st);
    else
        rips_sweep<2, W>(P, E, Ev, ord, psi, kf, h0, out.h0_cap, h1, out.h1_cap, k0, k1, st);
    PROF_MARK(3);
    PROF_COUNT(8, 1);
    PROF_COUNT(9, E);
    if (lane == 0) { out.h0_cnt[win] = k0; out.h1_cnt[win] = k1; out.status[win] = st; }
}

// ---------------------------------------------------------------------------------
// H1 rows -> ripser's order (descending birth; ties: descending death, then emission order)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
h1_order_kernel(double* __restrict__ h1, int h1_cap, const int* __restrict__ h1_cnt, int n_win)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* buf = reinterpret_cast<double*>(smem);
    const int win = blockIdx.x;
    if (win >= n_win) return;
    const int lane = lane_id();
    int k = h1_cnt[win];
    k = k < h1_cap ? k : h1_cap;
    if (k < 2) return;
    double* rows = h1 + (size_t)win * h1_cap * 2;
    for (int i = lane; i < 2 * k; i += 64) buf[i] = rows[i];
    WAVE_SYNC();
    for (int i = lane; i < k; i += 64) {
        const double bi = buf[2 * i], di = buf[2 * i + 1];
        int rank = 0;
        for (int j = 0; j < k; ++j) {
            const double bj = buf[2 * j], dj = buf[2 * j + 1];
            const bool before = (bj > bi) || (bj == bi && (dj > di || (dj == di && j < i)));
            rank += before ? 1 : 0;
        }
        rows[2 * rank] = bi;
        rows[2 * rank + 1] = di;
    }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static inline int align16(int x) { return (x + 15) & ~15; }

static tda_status order_h1(tda_ctx* ctx, double* h1, int h1_cap, int* h1_cnt, int n_win, hipStream_t st)
{
    if (h1_cap < 2) return TDA_OK;
    const size_t lds = (size_t)h1_cap * 16;
    if (lds > 64 * 1024) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "h1_cap too large (max 4096 rows)");
    hipLaunchKernelGGL(h1_order_kernel, dim3(n_win), dim3(64), lds, st, h1, h1_cap, h1_cnt, n_win);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

template <int NVW, int W>
static tda_status launch_dm_t(tda_ctx* ctx, const double* dm, int n_win, int n, float thresh, int symmetrise,
                              RipsOut out, hipStream_t st)
{
    const int E = n * (n - 1) / 2;
    int npad = 64;
    while (npad < E) npad <<= 1;
    const int s_bytes = npad * 8;
    const int psi_bytes = E * W * 8;
    const int off_ord = align16(s_bytes > psi_bytes ? s_bytes : psi_bytes);
    const int off_key = align16(off_ord + E * 2);
    const int total = align16(off_key + E * 4);
    auto kern = rips_dm_kernel<NVW, W>;
    if (total > 48 * 1024)
        TDA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, total));
    hipLaunchKernelGGL(kern, dim3(n_win), dim3(64), total, st, dm, n_win, n, thresh, symmetrise, off_ord, off_key,
                       out);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

tda_status launch_rips_dm(tda_ctx* ctx, const double* dm, int n_win, int n, double thresh, int symmetrise,
                          double* h0, int h0_cap, int* h0_cnt, double* h1, int h1_cap, int* h1_cnt, int* status,
                          hipStream_t st)
{
    if (n_win == 0) return TDA_OK;
    if (n < 1 || n > TDA_MAX_POINTS) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "n must be in [1,128]");
    if (h0_cap < n) TDA_FAIL(ctx, TDA_ERR_INVALID, "h0_cap must be >= n");
    RipsOut out{h0, h0_cap, h0_cnt, h1, h1_cap, h1_cnt, status};
    const float th = (float)thresh;
    tda_status rc;
    const int W = ctx->words_dm;
    if (n <= 64) {
        if (W == 1) rc = launch_dm_t<1, 1>(ctx, dm, n_win, n, th, symmetrise, out, st);
        else if (W == 2) rc = launch_dm_t<1, 2>(ctx, dm, n_win, n, th, symmetrise, out, st);
        else rc = launch_dm_t<1, 4>(ctx, dm, n_win, n, th, symmetrise, out, st);
    } else {
        // 128 classes only while [psi | ord | skey] still fits the 160 KiB LDS (n <= 112)
        const int E = n * (n - 1) / 2;
        const bool fits2 = (size_t)E * 16 + (size_t)E * 6 + 64 <= 160 * 1024;
        if (W == 1 || !fits2) rc = launch_dm_t<2, 1>(ctx, dm, n_win, n, th, symmetrise, out, st);
        else rc = launch_dm_t<2, 2>(ctx, dm, n_win, n, th, symmetrise, out, st);
    }
    if (rc != TDA_OK) return rc;
    return order_h1(ctx, h1, h1_cap, h1_cnt, n_win, st);
}

template <int W>
static tda_status launch_cloud_t(tda_ctx* ctx, const double* src, const int* aux, int n_win, int n_t_or_pcap,
                                 int dim, int subsample, int mode, int normalise, float thresh, int p_max,
                                 int* n_points, RipsOut out, hipStream_t st)
{
    const int E = p_max * (p_max - 1) / 2;
    int npad = 64;
    while (npad < E) npad <<= 1;
    const int off_psi = align16(E * 2);
    const int psi_end = off_psi + E * W * 8;
    const int s_bytes = npad * 8;
    const int off_pts = align16(psi_end > s_bytes ? psi_end : s_bytes);
    const int total = align16(off_pts + p_max * dim * 8);
    if (total > 160 * 1024) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "point cloud too large for LDS");
    auto kern = rips_cloud_kernel<W>;
    if (total > 48 * 1024)
        TDA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, total));
    hipLaunchKernelGGL(kern, dim3(n_win), dim3(64), total, st, src, aux, n_win, n_t_or_pcap, dim, subsample, mode,
                       normalise, thresh, off_psi, off_pts, p_max, n_points, out);
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
    if (p_max > TDA_MAX_POINTS) p_max = TDA_MAX_POINTS;   // larger windows are flagged per window
    if (p_max < 3) p_max = 3;
    if (h0_cap < p_max) TDA_FAIL(ctx, TDA_ERR_INVALID, "h0_cap must be >= max points per cloud");
    RipsOut out{h0, h0_cap, h0_cnt, h1, h1_cap, h1_cnt, status};
    const float th = (float)thresh;
    tda_status rc;
    if (ctx->words_cloud == 1)
        rc = launch_cloud_t<1>(ctx, src, aux, n_win, n_t_or_pcap, dim, subsample, mode, normalise, th, p_max,
                               n_points, out, st);
    else
        rc = launch_cloud_t<2>(ctx, src, aux, n_win, n_t_or_pcap, dim, subsample, mode, normalise, th, p_max,
                               n_points, out, st);
    if (rc != TDA_OK) return rc;
    return order_h1(ctx, h1, h1_cap, h1_cnt, n_win, st);
}

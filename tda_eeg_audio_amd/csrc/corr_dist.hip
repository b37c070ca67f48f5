// corr_dist.hip -- batched Pearson correlation -> distance, float64, gfx950.
//
// Replaces compute_correlation_matrix (np.corrcoef + NaN->0) and
// correlation_to_distance(method="euclidean") of notebooks/2_graph_construction.ipynb:86-122
// and the per-window Python loop of process_file_graphs (nb2:198-207).
//
// One (n_ch x n_t) window per 256-thread workgroup, three workgroups per CU.  The window is read from
// HBM once into registers (n_t <= 256; longer windows are streamed twice) and passes twice through a
// time-major LDS tile of CD_TCH samples x 16*NB channels:
//   pass A: channel means, one sequential sum over t per channel;
//   pass B: the tile is centred on its way into LDS and X X^T is accumulated on the f64 matrix cores:
//           v_mfma_f64_16x16x4_f64 over the upper-triangular 16x16 blocks, four samples per instruction.
// The instruction accumulates d = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c)))) -- measured on gfx950
// with tools/probes/mfma_f64_probe.hip: 0 mismatches in 51,200 chains of up to 252 terms -- so every
// (i,j) product is the ONE sequential fma chain over t = 0..n_t-1 that oracle/tda_oracle.c::orc_corr_dist
// fixes, and results are bit-identical to the oracle.  Zero padding (channels up to 16*NB, samples up to
// a multiple of 16) adds fma(0,0,acc) = acc.
// HBM-bound by design: 8*n_ch*n_t bytes in, 8*n_ch^2 (x2 with corr) bytes out per window.
#include "common.h"

#define CD_TCH 64         // time samples per LDS tile
#define CD_TSH 6          // log2(CD_TCH)
#define CD_MAXCH 64

typedef double d4 __attribute__((ext_vector_type(4)));

#ifdef TDA_PROFILE
__device__ unsigned long long g_prof_cd[16];
extern "C" __attribute__((visibility("default"))) int tda_profile_read_cd(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_cd), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof_cd), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
// phase times are accumulated in scalar registers and flushed once, so that the diagnostic build keeps the
// register allocation of the product build
#define CD_PROF_BEGIN() unsigned long long prof_t2 = clock64(), prof_a[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define CD_PROF_SUB(i)                                                        \
    do {                                                                      \
        const unsigned long long prof_t3 = clock64();                         \
        prof_a[i] += prof_t3 - prof_t2;                                       \
        prof_t2 = prof_t3;                                                    \
    } while (0)
#define CD_PROF_MARK(i) CD_PROF_SUB(i)
#define CD_PROF_END()                                                         \
    do {                                                                      \
        if (threadIdx.x == 0 && (blockIdx.x & 31) == 7) {                     \
            for (int i__ = 0; i__ < 14; ++i__) atomicAdd(&g_prof_cd[i__], prof_a[i__]); \
            atomicAdd(&g_prof_cd[14], 1ull);                                  \
        }                                                                     \
    } while (0)
#else
#define CD_PROF_BEGIN()
#define CD_PROF_MARK(i)
#define CD_PROF_SUB(i)
#define CD_PROF_END()
#endif

// global -> registers (issued early, consumed after the current tile has been used).  Element
// idx = tid + 256 k of a tile is (channel idx / 64, sample idx % 64): a wave reads 512 contiguous bytes.
template <int NPRE>
__device__ __forceinline__ void cd_fetch(double (&pre)[NPRE], const double* __restrict__ X, int ld, int n_ch, int c0,
                                         int tc, int tid)
{
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const int idx = tid + 256 * k;
        const int ch = idx >> CD_TSH, t = idx & (CD_TCH - 1);
        pre[k] = (ch < n_ch && t < tc) ? X[(size_t)ch * ld + c0 + t] : 0.0;
    }
}
// registers -> time-major LDS tile (row stride CSP = 16 NB + 1 doubles: conflict-free for this write,
// for the per-channel reads of pass A and for the MFMA operand reads), optionally centred
template <int NPRE, int CSP>
__device__ __forceinline__ void cd_stash(const double (&pre)[NPRE], double* tile, const double* mean, int n_ch, int tc,
                                         int tid, bool centre)
{
    double m[NPRE];                      // all means first: one LDS round trip instead of one per element
#pragma unroll
    for (int k = 0; k < NPRE; ++k) m[k] = centre ? mean[(tid + 256 * k) >> CD_TSH] : 0.0;   // mean[] is 0 beyond n_ch
    const int t = tid & (CD_TCH - 1);
    const bool live = t < tc;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const int ch = (tid + 256 * k) >> CD_TSH;
        tile[t * CSP + ch] = live ? pre[k] - m[k] : 0.0;      // x - 0 = x exactly
    }
}

#define CD_RES_CHUNKS 4   // windows of up to CD_RES_CHUNKS * CD_TCH samples stay in registers between the passes

// NB = ceil(n_ch / 16) row blocks.  RES: the whole window is fetched ONCE into registers (n_t <= 256, the
// reference's 250-sample windows); otherwise it is streamed twice, tile by tile, each tile in flight while the
// previous one is consumed.
template <int NB, bool RES>
__global__ void __launch_bounds__(256, NB == 4 ? 2 : 3)
corr_dist_kernel(const double* __restrict__ win, int n_win, int n_ch, int n_t, long long win_stride, int ld,
                 double* __restrict__ dist, double* __restrict__ corr)
{
    constexpr int CP = 16 * NB;                  // padded channel count
    constexpr int CSP = CP + 1;                  // tile row stride
    constexpr int NPRE = CP * CD_TCH / 256;      // tile elements per thread
    constexpr int NBLK = NB * (NB + 1) / 2;      // upper-triangular 16x16 blocks
    constexpr int MAXQ = (NBLK + 3) / 4;         // blocks per wave and tile
    constexpr int NBUF = RES ? CD_RES_CHUNKS : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);            // CD_TCH * CSP
    double* accs = tile + CD_TCH * CSP;                        // NBLK * 256: accumulators between tiles
    double* mean = accs + NBLK * 256;                          // CP
    double* sdev = mean + CP;                                  // CP
    const int w = blockIdx.x;
    if (w >= n_win) return;
    const int tid = threadIdx.x, l = tid & 63, wv = uni(tid >> 6);
    // window w = n_ch rows of n_t samples, row stride ld, starting win_stride elements after window w-1:
    //   stacked windows (preprocessed/<band>.npy): win_stride = n_ch*n_t, ld = n_t
    //   sliding windows over one band-passed recording (n_ch, L): win_stride = step, ld = L -- the 75 %
    //   overlap (nb1:338-341) is then served by L2 instead of being materialised 4x in HBM
    const double* X = win + (size_t)w * (size_t)win_stride;
    const int n_chunks = (n_t + CD_TCH - 1) / CD_TCH;
    double pre[NBUF][NPRE];
    CD_PROF_BEGIN();
    auto chunk_len = [&](int c) { const int r = n_t - c * CD_TCH; return r < CD_TCH ? r : CD_TCH; };
    // MFMA operand of block row r at sample t0: lane l holds x[16 r + (l & 15)][t0 + (l >> 4)]
    const double* lane_base = tile + (l >> 4) * CSP + (l & 15);

    // ---- pass A: channel means.  Wave r sums block row r with B = 1: fma(x, 1, s) = s + x correctly rounded,
    // i.e. the sequential sum over t, at 16 cycles per sample instead of the 44 of a dependent v_add_f64. ----
    d4 sacc = (d4){0.0, 0.0, 0.0, 0.0};
    auto sum_tile = [&](int tc) {
        if (wv < NB) {
            const double* p0 = lane_base + 16 * wv;
            const int ntr = (tc + 15) >> 4;                     // double trips of 2 x 8 samples; the tile is zero-padded
            double a[2][2];
            a[0][0] = p0[0]; a[0][1] = p0[4 * CSP];
            for (int d = 0; d < ntr; ++d) {
                const double* p1 = p0 + (16 * d + 8) * CSP;
                a[1][0] = p1[0]; a[1][1] = p1[4 * CSP];
                sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][0], 1.0, sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][1], 1.0, sacc, 0, 0, 0);
                const double* p2 = p0 + (16 * d + 16 < CD_TCH ? 16 * d + 16 : 0) * CSP;
                a[0][0] = p2[0]; a[0][1] = p2[4 * CSP];
                sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][0], 1.0, sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][1], 1.0, sacc, 0, 0, 0);
            }
        }
    };
    if constexpr (RES) {
#pragma unroll
        for (int c = 0; c < NBUF; ++c)
            if (c < n_chunks) cd_fetch<NPRE>(pre[c], X, ld, n_ch, c * CD_TCH, chunk_len(c), tid);
#pragma unroll
        for (int c = 0; c < NBUF; ++c)
            if (c < n_chunks) {
                CD_PROF_SUB(9);
                cd_stash<NPRE, CSP>(pre[c], tile, mean, n_ch, chunk_len(c), tid, false);
                CD_PROF_SUB(10);
                __syncthreads();
                CD_PROF_SUB(11);
                sum_tile(chunk_len(c));
                CD_PROF_SUB(12);
                __syncthreads();
                CD_PROF_SUB(13);
            }
    } else {
        cd_fetch<NPRE>(pre[0], X, ld, n_ch, 0, chunk_len(0), tid);
        for (int c = 0; c < n_chunks; ++c) {
            cd_stash<NPRE, CSP>(pre[0], tile, mean, n_ch, chunk_len(c), tid, false);
            __syncthreads();
            const int c1 = (c + 1 < n_chunks) ? c + 1 : 0;  // next tile of this pass, or the first tile of pass B
            cd_fetch<NPRE>(pre[0], X, ld, n_ch, c1 * CD_TCH, chunk_len(c1), tid);
            sum_tile(chunk_len(c));
            __syncthreads();
        }
    }
    // result layout of the instruction: register v of lane l is element (4 v + (l >> 4), l & 15) of the block
    if (wv < NB && (l & 15) == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int ch = 16 * wv + 4 * v + (l >> 4);
            mean[ch] = ch < n_ch ? sacc[v] / (double)n_t : 0.0;
        }
    }
    __syncthreads();
    CD_PROF_MARK(0);

    // ---- pass B: centred products on the matrix cores.  (block b, tile c) is work item c * NBLK + b and goes to
    // wave item % 4, so the waves (one per SIMD) carry equal loads; the accumulators wait in LDS between tiles. ----
    auto mfma_tile = [&](int c, int tc) {
        int br[MAXQ], bs[MAXQ], bb[MAXQ];
        d4 acc[MAXQ];
        const int b0 = (wv + 4 * NBLK - ((c * NBLK) & 3)) & 3;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            int b = b0 + 4 * q, r = 0;
            bb[q] = b;
            if (b >= NBLK) b = 0;                               // idle slot
            while (b >= NB - r) { b -= NB - r; ++r; }
            br[q] = r; bs[q] = r + b;
            if (bb[q] < NBLK && c > 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[q][v] = accs[(bb[q] * 4 + v) * 64 + l];
            } else acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
        }
        // trips of 8 samples (2 instructions per block); the operands of the next trip are in flight meanwhile
        double oa[2][MAXQ][2], ob[2][MAXQ][2];
        auto load_ops = [&](int trip, int s) {
#pragma unroll
            for (int q = 0; q < MAXQ; ++q)
                if (bb[q] < NBLK) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const double* p = lane_base + (8 * trip + 4 * u) * CSP;
                        oa[s][q][u] = p[16 * br[q]];
                        ob[s][q][u] = p[16 * bs[q]];
                    }
                }
        };
        auto run_ops = [&](int s) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < MAXQ; ++q)
                    if (bb[q] < NBLK) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[s][q][u], ob[s][q][u], acc[q], 0, 0, 0);
        };
        const int ntr = (tc + 15) >> 4;                          // double trips; the tile is zero-padded
        load_ops(0, 0);
        for (int d = 0; d < ntr; ++d) {
            load_ops(2 * d + 1, 1);
            run_ops(0);
            load_ops(2 * d + 2 < CD_TCH / 8 ? 2 * d + 2 : 0, 0);
            run_ops(1);
        }
#pragma unroll
        for (int q = 0; q < MAXQ; ++q)
            if (bb[q] < NBLK) {
#pragma unroll
                for (int v = 0; v < 4; ++v) accs[(bb[q] * 4 + v) * 64 + l] = acc[q][v];
            }
    };
    if constexpr (RES) {
#pragma unroll
        for (int c = 0; c < NBUF; ++c)
            if (c < n_chunks) {
                CD_PROF_SUB(3);
                cd_stash<NPRE, CSP>(pre[c], tile, mean, n_ch, chunk_len(c), tid, true);
                CD_PROF_SUB(4);
                __syncthreads();
                CD_PROF_SUB(5);
                mfma_tile(c, chunk_len(c));
                CD_PROF_SUB(6);
                __syncthreads();
                CD_PROF_SUB(7);
            }
    } else {
        for (int c = 0; c < n_chunks; ++c) {
            cd_stash<NPRE, CSP>(pre[0], tile, mean, n_ch, chunk_len(c), tid, true);
            __syncthreads();
            if (c + 1 < n_chunks) cd_fetch<NPRE>(pre[0], X, ld, n_ch, (c + 1) * CD_TCH, chunk_len(c + 1), tid);
            mfma_tile(c, chunk_len(c));
            __syncthreads();
        }
    }
    CD_PROF_MARK(1);

    // ---- covariance -> correlation -> distance.  Products commute, so the (j,i) chain is bit-identical to the
    // (i,j) chain and the lower triangle is read from the transposed position. ----
    const double fact = 1.0 / (double)(n_t - 1);
    auto cov_at = [&](int i, int j) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int r = lo >> 4, s = hi >> 4;
        const int b = r * NB - ((r * (r - 1)) >> 1) + (s - r);
        return accs[(b * 4 + ((lo & 15) >> 2)) * 64 + ((lo & 3) << 4) + (hi & 15)] * fact;
    };
    if (tid < n_ch) sdev[tid] = sqrt(cov_at(tid, tid));
    __syncthreads();
    double* Dw = dist + (size_t)w * n_ch * n_ch;
    double* Cw = corr ? corr + (size_t)w * n_ch * n_ch : nullptr;
    // one row per wave and trip (two in flight): lane = column
    for (int i = wv; i < n_ch; i += 8) {
        const int i2 = i + 4;
        const bool on = l < n_ch, on2 = on && i2 < n_ch;
        const int ii2 = i2 < n_ch ? i2 : i;
        const int j = on ? l : 0;
        double r1 = (cov_at(i, j) / sdev[i]) / sdev[j];
        double r2 = (cov_at(ii2, j) / sdev[ii2]) / sdev[j];
        r1 = r1 > 1.0 ? 1.0 : r1; r1 = r1 < -1.0 ? -1.0 : r1; if (r1 != r1) r1 = 0.0;    // clip; nan_to_num (nb2:95)
        r2 = r2 > 1.0 ? 1.0 : r2; r2 = r2 < -1.0 ? -1.0 : r2; if (r2 != r2) r2 = 0.0;
        double d1 = sqrt(2.0 * (1.0 - r1));        // nb2:108
        double d2 = sqrt(2.0 * (1.0 - r2));
        if (!(d1 > 0.0) || i == j) d1 = 0.0;       // nb2:119-120
        if (!(d2 > 0.0) || ii2 == j) d2 = 0.0;
        if (on) { if (Cw) Cw[i * n_ch + j] = r1; Dw[i * n_ch + j] = d1; }
        if (on2) { if (Cw) Cw[i2 * n_ch + j] = r2; Dw[i2 * n_ch + j] = d2; }
    }
    CD_PROF_MARK(2);
    CD_PROF_END();
}

// correlation_to_distance (nb2:100-122) on stored correlation matrices: all four methods.
// method: 0 "euclidean" sqrt(2(1-r)) | 1 "abs" 1-|r| | 2 "standard" 1-r | 3 "sqrt" sqrt(1-r^2)
__global__ void __launch_bounds__(256)
corr_to_dist_kernel(const double* __restrict__ corr, long long total, int n, int method, double* __restrict__ dist)
{
    const long long nn = (long long)n * n;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long e = idx % nn;
        const int i = (int)(e / n), j = (int)(e - (long long)i * n);
        double r = corr[idx];
        if (r > 1.0) r = 1.0;              // np.clip(corr, -1, 1), nb2:105 (NaN stays NaN)
        if (r < -1.0) r = -1.0;
        double d;
        if (method == 0) d = sqrt(2.0 * (1.0 - r));
        else if (method == 1) d = 1.0 - fabs(r);
        else if (method == 2) d = 1.0 - r;
        else d = sqrt(1.0 - r * r);
        if (d < 0.0) d = 0.0;              // np.maximum(d, 0), nb2:119 (NaN propagates like numpy)
        if (i == j) d = 0.0;               // nb2:120
        dist[idx] = d;
    }
}

tda_status launch_corr_to_dist(tda_ctx* ctx, const double* corr, int n_win, int n, int method, double* dist,
                               hipStream_t st)
{
    if (n_win == 0) return TDA_OK;
    if (method < 0 || method > 3) TDA_FAIL(ctx, TDA_ERR_INVALID, "Unknown method");
    const long long total = (long long)n_win * n * n;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(corr_to_dist_kernel, dim3((unsigned)blocks), dim3(256), 0, st, corr, total, n, method, dist);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

template <int NB>
static tda_status cd_launch_nb(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t, long long win_stride, int ld,
                               double* dist, double* corr, hipStream_t st)
{
    constexpr int CP = 16 * NB;
    const size_t lds = sizeof(double) * ((size_t)CD_TCH * (CP + 1) + (size_t)(NB * (NB + 1) / 2) * 256 + 2 * (size_t)CP);
    ProbeScope probe(ctx, TDA_PROBE_CORR_DIST, st);
    if (n_t <= CD_RES_CHUNKS * CD_TCH)
        hipLaunchKernelGGL((corr_dist_kernel<NB, true>), dim3(n_win), dim3(256), lds, st, win, n_win, n_ch, n_t, win_stride,
                           ld, dist, corr);
    else
        hipLaunchKernelGGL((corr_dist_kernel<NB, false>), dim3(n_win), dim3(256), lds, st, win, n_win, n_ch, n_t, win_stride,
                           ld, dist, corr);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}
static tda_status cd_launch(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t, long long win_stride, int ld,
                            double* dist, double* corr, hipStream_t st)
{
    switch ((n_ch + 15) / 16) {
    case 1: return cd_launch_nb<1>(ctx, win, n_win, n_ch, n_t, win_stride, ld, dist, corr, st);
    case 2: return cd_launch_nb<2>(ctx, win, n_win, n_ch, n_t, win_stride, ld, dist, corr, st);
    case 3: return cd_launch_nb<3>(ctx, win, n_win, n_ch, n_t, win_stride, ld, dist, corr, st);
    default: return cd_launch_nb<4>(ctx, win, n_win, n_ch, n_t, win_stride, ld, dist, corr, st);
    }
}

tda_status launch_corr_dist(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t, double* dist,
                            double* corr, hipStream_t st)
{
    if (n_win == 0) return TDA_OK;
    if (n_ch < 1 || n_ch > CD_MAXCH) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "n_ch must be in [1,64]");
    if (n_t < 2) TDA_FAIL(ctx, TDA_ERR_INVALID, "n_t must be >= 2");
    return cd_launch(ctx, win, n_win, n_ch, n_t, (long long)n_ch * n_t, n_t, dist, corr, st);
}

// sliding windows over a (n_ch, n_samples) band-passed recording: create_sliding_windows
// (notebooks/1_preprocesamiento.ipynb:314-381) fused with the per-window corr->dist loop (nb2:198-207)
tda_status launch_corr_dist_sliding(tda_ctx* ctx, const double* sig, int n_ch, int n_samples, int win_len, int step,
                                    double* dist, double* corr, int* n_win_out, hipStream_t st)
{
    if (n_ch < 1 || n_ch > CD_MAXCH) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "n_ch must be in [1,64]");
    if (win_len < 2 || step < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "win_len must be >= 2 and step >= 1");
    const int n_win = n_samples >= win_len ? (n_samples - win_len) / step + 1 : 0;   // nb1:341
    if (n_win_out) *n_win_out = n_win;
    if (n_win == 0) return TDA_OK;
    return cd_launch(ctx, sig, n_win, n_ch, win_len, (long long)step, n_samples, dist, corr, st);
}

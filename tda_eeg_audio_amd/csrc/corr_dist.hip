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

#include "corr_dist_dev.h"

// NB = ceil(n_ch / 16) row blocks.  RES: the whole window is fetched ONCE into registers (n_t <= 256, the
// reference's 250-sample windows); otherwise it is streamed twice, tile by tile, each tile in flight while the
// previous one is consumed.  (The two passes live in corr_dist_dev.h, shared with the fused EEG kernel of rips.hip.)
template <int NB, bool RES>
__global__ void __launch_bounds__(256, NB == 4 ? 2 : 3)
corr_dist_kernel(const double* __restrict__ win, int n_win, int n_ch, int n_t, long long win_stride, int ld,
                 double* __restrict__ dist, double* __restrict__ corr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = blockIdx.x;
    if (w >= n_win) return;
    const int tid = threadIdx.x, l = tid & 63, wv = uni(tid >> 6);
    // window w = n_ch rows of n_t samples, row stride ld, starting win_stride elements after window w-1:
    //   stacked windows (preprocessed/<band>.npy): win_stride = n_ch*n_t, ld = n_t
    //   sliding windows over one band-passed recording (n_ch, L): win_stride = step, ld = L -- the 75 %
    //   overlap (nb1:338-341) is then served by L2 instead of being materialised 4x in HBM
    cd_window_products<NB, RES>(smem, win + (size_t)w * (size_t)win_stride, n_ch, n_t, ld);
    const double* accs = reinterpret_cast<const double*>(smem) + CdLayout<NB>::TILE;
    const double* sdev = accs + CdLayout<NB>::ACCS + CdLayout<NB>::CP;

    // ---- covariance -> correlation -> distance.  Products commute, so the (j,i) chain is bit-identical to the
    // (i,j) chain and the lower triangle is read from the transposed position. ----
    const double fact = 1.0 / (double)(n_t - 1);
    auto cov_at = [&](int i, int j) { return accs[CdLayout<NB>::acc_index(i, j)] * fact; };
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
    const size_t lds = CdLayout<NB>::BYTES;
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

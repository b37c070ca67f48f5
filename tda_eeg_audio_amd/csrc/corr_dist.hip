// corr_dist.hip -- batched Pearson correlation -> distance, float64, gfx950.
//
// Replaces compute_correlation_matrix (np.corrcoef + NaN->0) and
// correlation_to_distance(method="euclidean") of notebooks/2_graph_construction.ipynb:86-122
// and the per-window Python loop of process_file_graphs (nb2:198-207).
//
// One (n_ch x n_t) window per 256-thread workgroup.  The window is streamed twice through a
// (n_ch x TC) LDS tile (second pass is an L2 hit): pass A accumulates the channel means,
// pass B centres on the way into LDS and accumulates 3x3 register tiles of X X^T.
// Every (i,j) product is ONE sequential fma chain over t = 0..n_t-1 -- the operation order
// oracle/tda_oracle.c::orc_corr_dist fixes -- so results are bit-identical to the oracle.
// HBM-bound by design: 8*n_ch*n_t bytes in, 8*n_ch^2 (x2 with corr) bytes out per window.
#include "common.h"

#define CD_TC 50          // time samples per LDS tile
#define CD_TCP (CD_TC + 1)
#define CD_MAXCH 64

__global__ void __launch_bounds__(256)
corr_dist_kernel(const double* __restrict__ win, int n_win, int n_ch, int n_t, long long win_stride, int ld,
                 double* __restrict__ dist, double* __restrict__ corr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);            // n_ch * CD_TCP
    double* cmat = tile + n_ch * CD_TCP;                       // n_ch * n_ch
    double* mean = cmat + n_ch * n_ch;                         // n_ch
    double* sdev = mean + n_ch;                                // n_ch
    const int w = blockIdx.x;
    if (w >= n_win) return;
    const int tid = threadIdx.x;
    // window w = n_ch rows of n_t samples, row stride ld, starting win_stride elements after window w-1:
    //   stacked windows (preprocessed/<band>.npy): win_stride = n_ch*n_t, ld = n_t
    //   sliding windows over one band-passed recording (n_ch, L): win_stride = step, ld = L -- the 75 %
    //   overlap (nb1:338-341) is then served by L2 instead of being materialised 4x in HBM
    const double* X = win + (size_t)w * (size_t)win_stride;

    // 3x3 tile owned by this thread: (ti <= tj) over a T x T tile grid
    const int T = (n_ch + 2) / 3;
    int ti = 0, rem = tid;
    while (ti < T && rem >= T - ti) { rem -= T - ti; ++ti; }
    const bool has_tile = ti < T;
    const int tj = ti + rem;
    const int i0 = 3 * ti, j0 = 3 * tj;

    // ---- pass A: channel means (sequential sum over t, one thread per channel) ----
    double msum = 0.0;
    for (int c0 = 0; c0 < n_t; c0 += CD_TC) {
        const int tc = (n_t - c0) < CD_TC ? (n_t - c0) : CD_TC;
        for (int idx = tid; idx < n_ch * tc; idx += 256) {
            const int ch = idx / tc, t = idx - ch * tc;
            tile[ch * CD_TCP + t] = X[(size_t)ch * ld + c0 + t];
        }
        __syncthreads();
        if (tid < n_ch)
            for (int t = 0; t < tc; ++t) msum += tile[tid * CD_TCP + t];
        __syncthreads();
    }
    if (tid < n_ch) mean[tid] = msum / (double)n_t;
    __syncthreads();

    // ---- pass B: centred products ----
    double acc[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) acc[u][v] = 0.0;
    for (int c0 = 0; c0 < n_t; c0 += CD_TC) {
        const int tc = (n_t - c0) < CD_TC ? (n_t - c0) : CD_TC;
        for (int idx = tid; idx < n_ch * tc; idx += 256) {
            const int ch = idx / tc, t = idx - ch * tc;
            tile[ch * CD_TCP + t] = X[(size_t)ch * ld + c0 + t] - mean[ch];
        }
        __syncthreads();
        if (has_tile) {
            // rows beyond n_ch read row 0 (results discarded)
            const int ri0 = (i0 < n_ch ? i0 : 0) * CD_TCP, ri1 = (i0 + 1 < n_ch ? i0 + 1 : 0) * CD_TCP,
                      ri2 = (i0 + 2 < n_ch ? i0 + 2 : 0) * CD_TCP;
            const int rj0 = (j0 < n_ch ? j0 : 0) * CD_TCP, rj1 = (j0 + 1 < n_ch ? j0 + 1 : 0) * CD_TCP,
                      rj2 = (j0 + 2 < n_ch ? j0 + 2 : 0) * CD_TCP;
            for (int t = 0; t < tc; ++t) {
                const double a0 = tile[ri0 + t], a1 = tile[ri1 + t], a2 = tile[ri2 + t];
                const double b0 = tile[rj0 + t], b1 = tile[rj1 + t], b2 = tile[rj2 + t];
                acc[0][0] = fma(a0, b0, acc[0][0]); acc[0][1] = fma(a0, b1, acc[0][1]); acc[0][2] = fma(a0, b2, acc[0][2]);
                acc[1][0] = fma(a1, b0, acc[1][0]); acc[1][1] = fma(a1, b1, acc[1][1]); acc[1][2] = fma(a1, b2, acc[1][2]);
                acc[2][0] = fma(a2, b0, acc[2][0]); acc[2][1] = fma(a2, b1, acc[2][1]); acc[2][2] = fma(a2, b2, acc[2][2]);
            }
        }
        __syncthreads();
    }
    const double fact = 1.0 / (double)(n_t - 1);
    if (has_tile) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int i = i0 + u, j = j0 + v;
                if (i < n_ch && j < n_ch) {
                    const double c = acc[u][v] * fact;
                    // products commute, so the (j,i) chain is bit-identical to the (i,j) chain
                    cmat[i * n_ch + j] = c;
                    cmat[j * n_ch + i] = c;
                }
            }
    }
    __syncthreads();
    if (tid < n_ch) sdev[tid] = sqrt(cmat[tid * n_ch + tid]);
    __syncthreads();
    double* Dw = dist + (size_t)w * n_ch * n_ch;
    double* Cw = corr ? corr + (size_t)w * n_ch * n_ch : nullptr;
    for (int idx = tid; idx < n_ch * n_ch; idx += 256) {
        const int i = idx / n_ch, j = idx - i * n_ch;
        double r = (cmat[i * n_ch + j] / sdev[i]) / sdev[j];
        if (r > 1.0) r = 1.0;
        if (r < -1.0) r = -1.0;
        if (r != r) r = 0.0;                       // nan_to_num (nb2:95)
        if (Cw) Cw[idx] = r;
        double d = sqrt(2.0 * (1.0 - r));          // nb2:108
        if (!(d > 0.0)) d = 0.0;                   // nb2:119
        if (i == j) d = 0.0;                       // nb2:120
        Dw[idx] = d;
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

tda_status launch_corr_dist(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t, double* dist,
                            double* corr, hipStream_t st)
{
    if (n_win == 0) return TDA_OK;
    if (n_ch < 1 || n_ch > CD_MAXCH) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "n_ch must be in [1,64]");
    if (n_t < 2) TDA_FAIL(ctx, TDA_ERR_INVALID, "n_t must be >= 2");
    const size_t lds = sizeof(double) * ((size_t)n_ch * CD_TCP + (size_t)n_ch * n_ch + 2 * (size_t)n_ch);
    hipLaunchKernelGGL(corr_dist_kernel, dim3(n_win), dim3(256), lds, st, win, n_win, n_ch, n_t,
                       (long long)n_ch * n_t, n_t, dist, corr);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
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
    const size_t lds = sizeof(double) * ((size_t)n_ch * CD_TCP + (size_t)n_ch * n_ch + 2 * (size_t)n_ch);
    hipLaunchKernelGGL(corr_dist_kernel, dim3(n_win), dim3(256), lds, st, sig, n_win, n_ch, win_len, (long long)step,
                       n_samples, dist, corr);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

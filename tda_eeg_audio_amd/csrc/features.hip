// features.hip -- small per-window kernels around the Rips core, gfx950 / wave64.
//
//   tau_kernel        replaces compute_tau                  (scripts/utils.py:92-104)
//   features_kernel   replaces extract_features             (scripts/utils.py:144-177)
//                     == extract_persistence_features       (tda_eeg_classification_v2.py:179-250)
//   aggregate_kernel  replaces the mean/std over windows    (tda_eeg_classification_v2.py:429-436)
//
// All float64.  Sums follow numpy's pairwise-summation tree (8 interleaved partial sums for
// n <= 128, halving above) so that mean/std agree with np.mean/np.std to the last bit on the
// same input order; log() is OCML's, so the entropy is compared with a 1e-12 tolerance.
#include "common.h"
#include <climits>

// numpy's pairwise sum over f(lo) .. f(lo+n-1), evaluated redundantly by every lane that
// calls it (n is tiny: <= a few hundred).  The leaf (n <= 128: every diagram and group of the path) is inlined into
// its caller -- as a (recursive) call per sum the finishing pass of a step took 1.6x as long.
template <class F>
__device__ __forceinline__ double np_pairwise_leaf(const F& f, int lo, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += f(lo + i);
        return res;
    }
    double r0 = f(lo), r1 = f(lo + 1), r2 = f(lo + 2), r3 = f(lo + 3);
    double r4 = f(lo + 4), r5 = f(lo + 5), r6 = f(lo + 6), r7 = f(lo + 7);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += f(lo + i + 0); r1 += f(lo + i + 1); r2 += f(lo + i + 2); r3 += f(lo + i + 3);
        r4 += f(lo + i + 4); r5 += f(lo + i + 5); r6 += f(lo + i + 6); r7 += f(lo + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += f(lo + i);
    return res;
}
template <class F>
__device__ __forceinline__ double np_pairwise_split_uniform(const F& f, int lo_, int n_)        // n > 128 (off the usual path)
{
    // numpy halves recursively (left part a multiple of 8) down to leaves of <= 128: the same tree, walked with an
    // explicit stack.  UNIFORM flavour: every lane of the wave evaluates the same sum, so frame i of the stack lives in LANE i of four
    // registers (select / v_readlane with the wave-uniform stack pointer): no calls, no scratch memory.
    const int lo = uni(lo_), n = uni(n_);
    int vlo = 0, vn = 0, vph = 0, vl0 = 0, vl1 = 0;
    int sp = 0;
    const int me = lane_id();
    double ret = 0.0;
    vlo = me == 0 ? (lo) : vlo;
    vn = me == 0 ? (n) : vn;
    while (sp >= 0) {
        const int flo = __builtin_amdgcn_readlane(vlo, sp), fn = __builtin_amdgcn_readlane(vn, sp);
        const int ph = __builtin_amdgcn_readlane(vph, sp);
        if (fn <= 128) { ret = uni_f64(np_pairwise_leaf(f, flo, fn), 0); --sp; continue; }
        int n2 = fn / 2;
        n2 -= n2 % 8;
        if (ph == 0) {
            vph = me == sp ? (1) : vph;
            ++sp;
            vlo = me == sp ? (flo) : vlo; vn = me == sp ? (n2) : vn;
            vph = me == sp ? (0) : vph;
        } else if (ph == 1) {
            const long long bits = __double_as_longlong(ret);
            vl0 = me == sp ? ((int)(unsigned)(bits & 0xffffffffll)) : vl0;
            vl1 = me == sp ? ((int)(unsigned)((unsigned long long)bits >> 32)) : vl1;
            vph = me == sp ? (2) : vph;
            ++sp;
            vlo = me == sp ? (flo + n2) : vlo; vn = me == sp ? (fn - n2) : vn;
            vph = me == sp ? (0) : vph;
        } else {
            const unsigned l0 = (unsigned)__builtin_amdgcn_readlane(vl0, sp), l1 = (unsigned)__builtin_amdgcn_readlane(vl1, sp);
            ret = __longlong_as_double((long long)(((unsigned long long)l1 << 32) | l0)) + ret;
            --sp;
        }
    }
    return ret;
}
// any-lane flavour (sums and lengths may differ from lane to lane): frames in private arrays, a real call -- used by
// the small per-group kernels, whose groups have <= 128 members on the whole path
template <class F>
__device__ __noinline__ double np_pairwise_split(const F& f, int lo, int n)
{
    int slo[26], sn[26], sph[26];
    double sleft[26];
    int sp = 0;
    double ret = 0.0;
    slo[0] = lo; sn[0] = n; sph[0] = 0; sleft[0] = 0.0;
    while (sp >= 0) {
        const int flo = slo[sp], fn = sn[sp];
        if (fn <= 128) { ret = np_pairwise_leaf(f, flo, fn); --sp; continue; }
        int n2 = fn / 2;
        n2 -= n2 % 8;
        if (sph[sp] == 0) { sph[sp] = 1; ++sp; slo[sp] = flo; sn[sp] = n2; sph[sp] = 0; }
        else if (sph[sp] == 1) { sleft[sp] = ret; sph[sp] = 2; ++sp; slo[sp] = flo + n2; sn[sp] = fn - n2; sph[sp] = 0; }
        else { ret = sleft[sp] + ret; --sp; }
    }
    return ret;
}
template <class F>
__device__ __forceinline__ double np_pairwise_fn(const F& f, int lo, int n)
{
    return n <= 128 ? np_pairwise_leaf(f, lo, n) : np_pairwise_split(f, lo, n);
}

__device__ __forceinline__ double np_pairwise_sum(const double* a, int n, int stride)
{
    auto f = [=](int i) { return a[(size_t)i * stride]; };
    return np_pairwise_fn(f, 0, n);
}
// the same sum and length in every lane of the wave (diagram_finish_kernel)
__device__ __forceinline__ double np_pairwise_sum_uniform(const double* a, int n)
{
    auto f = [=](int i) { return a[i]; };
    return n <= 128 ? np_pairwise_leaf(f, 0, n) : np_pairwise_split_uniform(f, 0, n);
}

// ---------------------------------------------------------------------------------
// compute_tau: first lag k in [1, min(max_lag, len)) whose autocorrelation is <= 0.
// One wave per window; lane l owns lags l+1 and l+65 (max_lag <= 128), each a
// sequential fma chain over t (the order oracle/tda_oracle.c::orc_compute_tau fixes).
// ---------------------------------------------------------------------------------
// seg_off != nullptr: workgroup g handles the FIRST window of group g (tda_eeg_audio_comparison.py:83: tau is
// computed once per recording-band, from its first selected window) and also writes the value to every
// window of the group in tau_win, the per-window array the Takens kernel takes.
__global__ void __launch_bounds__(64)
tau_kernel(const double* __restrict__ win, int n_win, int n_t, int max_lag, int* __restrict__ tau,
           const int* __restrict__ seg_off, int* __restrict__ tau_win)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* sc = reinterpret_cast<double*>(smem);
    const int g = blockIdx.x;
    if (g >= n_win) return;                          // n_win = number of groups in segment mode
    const int lane = lane_id();
    int w = g, w_end = g + 1;
    if (seg_off) {
        w = seg_off[g]; w_end = seg_off[g + 1];
        if (w_end <= w) { if (lane == 0) tau[g] = 0; return; }      // empty group
    }
    const double* s = win + (size_t)w * n_t;
    for (int t = lane; t < n_t; t += 64) sc[t] = s[t];
    __syncthreads();
    double sum = 0.0;
    for (int t = 0; t < n_t; ++t) sum += sc[t];     // every lane: same sequential sum
    const double m = sum / (double)n_t;
    __syncthreads();
    for (int t = lane; t < n_t; t += 64) sc[t] = sc[t] - m;
    __syncthreads();
    if (max_lag < 0) max_lag = n_t / 4;             // utils.py:94-95
    if (max_lag > n_t - 1) max_lag = n_t - 1;       // utils.py:96
    const int lim = max_lag < n_t ? max_lag : n_t;  // utils.py:101
    int found = 0;
    for (int k0 = 1; k0 < lim && !found; k0 += 64) {
        const int k = k0 + lane;
        double acc = 1.0;
        if (k < lim) {
            acc = 0.0;
            for (int t = 0; t + k < n_t; ++t) acc = fma(sc[t + k], sc[t], acc);
        }
        const u64 bal = __ballot(k < lim && acc <= 0.0);
        if (bal) found = k0 + __builtin_ctzll(bal);
    }
    if (!found) { found = max_lag / 10; if (found < 1) found = 1; }
    if (lane == 0) tau[g] = found;
    if (tau_win)
        for (int i = w + lane; i < w_end; i += 64) tau_win[i] = found;
}

// ---------------------------------------------------------------------------------
// Finishing pass over the diagrams of a batch, up to four diagram sets in ONE launch, one wave per diagram:
//   * order != 0: the rows of an H1 diagram go into ripser's order (descending birth; ties: descending death, then
//     emission order) -- the Rips kernels emit them in the order of the kills;
//   * feat != NULL: extract_features, 11 scalars per diagram, key order of utils.py:166-177.
// This is latency-bound scalar work (a few dependent float64 sums over <= a few hundred rows): it wants many
// independent waves and little LDS each, which is why it is NOT an epilogue of the Rips kernels (an 80 KB
// workgroup would sit on its CU for the duration; measured: 13 % slower end to end).
// ---------------------------------------------------------------------------------
struct DiagramSets {
    double* rows[4]; const int* cnt[4]; int cap[4]; int order[4]; double* feat[4];
    int n_sets;
};

// Waves of a workgroup never meet: each has its own diagram and its own LDS slice, and the LDS serves the accesses
// of ONE wave in program order -- a compiler fence is all the synchronisation there is.  (Four diagrams per
// workgroup because a launch of 3 x 106,200 one-wave workgroups is bound by the dispatcher, not by the work.)
#define FIN_WAVES 4
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// A launch takes the diagrams with k_lo < rows <= lds_cap: the first one has slices of 64 rows (8 KB per workgroup:
// eight workgroups = every wave slot of a CU), a second one, sized by the capacities of the buffers (H1: 256 rows,
// 32 KB per workgroup, five per CU), the larger diagrams -- its other waves leave after one load.
__global__ void __launch_bounds__(64 * FIN_WAVES, 8)
diagram_finish_kernel(DiagramSets S, int n_dgm, int lds_cap, int k_lo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wv = uni((int)(threadIdx.x >> 6));
    double* b = reinterpret_cast<double*>(smem) + (size_t)wv * 4 * lds_cap;    // births | deaths | pers | tmp, lds_cap each
    double* d = b + lds_cap;
    double* p = d + lds_cap;
    double* tmp = p + lds_cap;
    const long long gi = (long long)blockIdx.x * (blockDim.x >> 6) + wv;
    const int set = (int)(gi / n_dgm), g = (int)(gi - (long long)set * n_dgm);
    if (set >= S.n_sets) return;
    const int cap = S.cap[set];
    const int lane = lane_id();
    int k = S.cnt[set][g];
    k = k < cap ? k : cap;
    if (k > lds_cap || k <= k_lo) return;              // (wave-uniform: another launch's diagram)
    double* rows = S.rows[set] + (size_t)g * cap * 2;
    double* feat = S.feat[set];
    const bool reorder = S.order[set] && k >= 2;
    if (!reorder && !feat) return;
    // rows -> LDS (p, tmp serve as staging while the rows are put in order)
    double* rb = reorder ? p : b;
    double* rd = reorder ? tmp : d;
    for (int i = lane; i < k; i += 64) { rb[i] = rows[2 * i]; rd[i] = rows[2 * i + 1]; }
    wave_sync();
    if (reorder) {
        for (int i = lane; i < k; i += 64) {
            const double bi = rb[i], di = rd[i];
            int pos = 0;
            for (int j = 0; j < k; ++j) {
                const double bj = rb[j], dj = rd[j];
                pos += ((bj > bi) || (bj == bi && (dj > di || (dj == di && j < i)))) ? 1 : 0;
            }
            b[pos] = bi; d[pos] = di;
            if (pos != i) { rows[2 * pos] = bi; rows[2 * pos + 1] = di; }
        }
        wave_sync();
    }
    if (!feat) return;
    // compact finite rows in place, preserving order (utils.py:146-147): row i moves to pos <= i, and every lane
    // has read its row before any lane of the same trip writes
    int m = 0, ness = 0;
    for (int i0 = 0; i0 < k; i0 += 64) {
        const int i = i0 + lane;
        double bi = 0.0, di = 0.0;
        bool fin = false, valid = i < k;
        if (valid) { bi = b[i]; di = d[i]; fin = isfinite(bi) && isfinite(di); }
        const u64 bal = __ballot(fin);
        const int pos = m + __popcll(bal & ((1ull << lane) - 1ull));
        wave_sync();
        if (fin) { b[pos] = bi; d[pos] = di; p[pos] = di - bi; }
        m += __popcll(bal);
        ness += __popcll(__ballot(valid && !fin));
    }
    wave_sync();
    double out[TDA_N_FEATURES];
#pragma unroll
    for (int i = 0; i < TDA_N_FEATURES; ++i) out[i] = 0.0;
    out[1] = (double)ness;
    if (m > 0) {
        out[0] = (double)m;
        const double* arr[3] = {b, d, p};
        for (int q = 0; q < 3; ++q) {
            const double mean = np_pairwise_sum_uniform(arr[q], m) / (double)m;
            out[2 + 2 * q] = mean;
            if (m > 1) {
                wave_sync();
                for (int i = lane; i < m; i += 64) { const double z = arr[q][i] - mean; tmp[i] = z * z; }
                wave_sync();
                out[3 + 2 * q] = sqrt(np_pairwise_sum_uniform(tmp, m) / (double)m);
            }
        }
        double mx = -INFINITY;
        for (int i = lane; i < m; i += 64) mx = p[i] > mx ? p[i] : mx;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
        out[8] = mx;
        const double tot = np_pairwise_sum_uniform(p, m);
        out[9] = tot;
        if (m > 1 && tot > 0.0) {
            wave_sync();
            // pn = pers/sum; keep pn > 0 in order (utils.py:161-163)
            int c = 0;
            for (int i0 = 0; i0 < m; i0 += 64) {
                const int i = i0 + lane;
                double pn = 0.0;
                if (i < m) pn = p[i] / tot;
                const bool pos = i < m && pn > 0.0;
                const u64 bal = __ballot(pos);
                const int at = c + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos) tmp[at] = pn * log(pn + 1e-10);
                c += __popcll(bal);
            }
            wave_sync();
            out[10] = -np_pairwise_sum_uniform(tmp, c) / log((double)m + 1e-10);
        }
    }
    if (lane < TDA_N_FEATURES) {
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < TDA_N_FEATURES; ++i) v = (lane == i) ? out[i] : v;
        feat[(size_t)g * TDA_N_FEATURES + lane] = v;
    }
}

// ---------------------------------------------------------------------------------
// per (recording, band) aggregation: np.mean / np.std over the used windows
// out[seg][f*4 + {0,1,2,3}] = h0 mean, h0 std, h1 mean, h1 std   (v2:429-436)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
aggregate_kernel(const double* __restrict__ f0, const double* __restrict__ f1, const int* __restrict__ seg_off,
                 int n_seg, double* __restrict__ out)
{
    const int seg = blockIdx.x;
    if (seg >= n_seg) return;
    const int lane = lane_id();
    if (lane >= 2 * TDA_N_FEATURES) return;
    const int h = lane / TDA_N_FEATURES, f = lane % TDA_N_FEATURES;
    const int s0 = seg_off[seg], s1 = seg_off[seg + 1];
    const int n = s1 - s0;
    const double* src = (h == 0 ? f0 : f1) + (size_t)s0 * TDA_N_FEATURES + f;
    double mean = 0.0, sd = 0.0;
    if (n > 0) {
        mean = np_pairwise_sum(src, n, TDA_N_FEATURES) / (double)n;
        // np.std = sqrt(mean(|x-mean|^2)), same pairwise tree evaluated on the fly
        auto sq = [=](int i) { const double z = src[(size_t)i * TDA_N_FEATURES] - mean; return z * z; };
        const double res = np_pairwise_fn(sq, 0, n);
        sd = sqrt(res / (double)n);
    }
    double* o = out + (size_t)seg * (4 * TDA_N_FEATURES) + f * 4 + h * 2;
    o[0] = mean;
    o[1] = sd;
}

// ---------------------------------------------------------------------------------
// np.nanmean over the windows of each (recording, band) group  (cmp:117-118, mvm:95)
// one lane per segment; sums follow numpy's pairwise tree over the non-NaN values
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
nanmean_kernel(const double* __restrict__ x, const int* __restrict__ seg_off, int n_seg, double* __restrict__ out)
{
    const int seg = blockIdx.x * 64 + threadIdx.x;
    if (seg >= n_seg) return;
    const int s0 = seg_off[seg], s1 = seg_off[seg + 1];
    // np.nanmean: NaNs replaced by 0, sum / count of non-NaN; all-NaN (or empty) -> NaN
    int cnt = 0;
    for (int i = s0; i < s1; ++i) cnt += (x[i] == x[i]) ? 1 : 0;
    auto val = [=](int i) { const double v = x[s0 + i]; return v == v ? v : 0.0; };
    const double sum = np_pairwise_fn(val, 0, s1 - s0);
    out[seg] = cnt > 0 ? sum / (double)cnt : __longlong_as_double(0x7ff8000000000000ll);
}

// ---------------------------------------------------------------------------------
// One row per (recording, band) group, the unit the GPUs exchange:
//   [ nanmean W_H0 (cmp:117), nanmean W_H1 (cmp:118), tau (cmp:83), n_windows, 44 aggregated EEG features (v2:429-436) ]
// = nanmean_kernel x 2 + aggregate_kernel + the row assembly in one launch (lanes 0..21 aggregate, 22/23 nanmean).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
recording_rows_kernel(const double* __restrict__ w0, const double* __restrict__ w1, const int* __restrict__ tau_seg,
                      const double* __restrict__ f0, const double* __restrict__ f1, const int* __restrict__ seg_off,
                      int n_seg, double* __restrict__ out, const int* __restrict__ status_a,
                      const int* __restrict__ status_b, int* __restrict__ seg_flags)
{
    const int seg = blockIdx.x;
    if (seg >= n_seg) return;
    const int lane = lane_id();
    const int s0 = seg_off[seg], s1 = seg_off[seg + 1];
    const int n = s1 - s0;
    double* row = out + (size_t)seg * (4 + 4 * TDA_N_FEATURES);
    if (lane < 2 * TDA_N_FEATURES) {
        const int h = lane / TDA_N_FEATURES, f = lane % TDA_N_FEATURES;
        const double* src = (h == 0 ? f0 : f1) + (size_t)s0 * TDA_N_FEATURES + f;
        double mean = 0.0, sd = 0.0;
        if (n > 0) {
            mean = np_pairwise_sum(src, n, TDA_N_FEATURES) / (double)n;
            auto sq = [=](int i) { const double z = src[(size_t)i * TDA_N_FEATURES] - mean; return z * z; };
            sd = sqrt(np_pairwise_fn(sq, 0, n) / (double)n);
        }
        row[4 + f * 4 + h * 2] = mean;
        row[4 + f * 4 + h * 2 + 1] = sd;
    } else if (lane < 2 * TDA_N_FEATURES + 2) {
        // cmp:90-91: a window whose Takens cloud has fewer than 3 points (or none at all) never reaches the
        // distances; np.nanmean then runs over the list of the windows that did (cmp:117-118).  No window left:
        // the reference drops the band (cmp:101-102), here the two distances are NaN.
        const double* x = lane == 2 * TDA_N_FEATURES ? w0 : w1;
        const int skip = TDA_WIN_DEGENERATE | TDA_WIN_TOO_LARGE;
        int m = 0, cnt = 0;
        for (int i = s0; i < s1; ++i) {
            if (status_b && (status_b[i] & skip)) continue;
            ++m;
            cnt += (x[i] == x[i]) ? 1 : 0;
        }
        // the j-th surviving window (the survivors keep their order, so the pairwise tree is numpy's)
        auto val = [=](int j) {
            int i = s0;
            if (status_b) { for (int seen = -1;; ++i) { if (!(status_b[i] & skip) && ++seen == j) break; } }
            else i = s0 + j;
            const double v = x[i];
            return v == v ? v : 0.0;
        };
        const double sum = np_pairwise_fn(val, 0, m);
        row[lane - 2 * TDA_N_FEATURES] = cnt > 0 ? sum / (double)cnt : __longlong_as_double(0x7ff8000000000000ll);
    } else if (lane == 2 * TDA_N_FEATURES + 2) {
        row[2] = (double)tau_seg[seg];
        row[3] = (double)n;
    } else if (lane == 2 * TDA_N_FEATURES + 3 && seg_flags) {
        int fl = 0;
        for (int i = s0; i < s1; ++i) fl |= (status_a ? status_a[i] : 0) | (status_b ? status_b[i] : 0);
        seg_flags[seg] = fl & ~TDA_WIN_DEGENERATE;          // every condition a caller has to act on (a degenerate cloud is a result)
    }
}

tda_status launch_recording_rows(tda_ctx* ctx, const double* w0, const double* w1, const int* tau_seg, const double* f0,
                                 const double* f1, const int* seg_off, int n_seg, double* out, const int* status_a,
                                 const int* status_b, int* seg_flags, hipStream_t st)
{
    if (n_seg == 0) return TDA_OK;
    hipLaunchKernelGGL(recording_rows_kernel, dim3(n_seg), dim3(64), 0, st, w0, w1, tau_seg, f0, f1, seg_off, n_seg, out,
                       status_a, status_b, seg_flags);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

tda_status launch_nanmean(tda_ctx* ctx, const double* x, const int* seg_off, int n_seg, double* out, hipStream_t st)
{
    if (n_seg == 0) return TDA_OK;
    hipLaunchKernelGGL(nanmean_kernel, dim3((n_seg + 63) / 64), dim3(64), 0, st, x, seg_off, n_seg, out);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

// ---------------------------------------------------------------------------------
// Spearman correlation of two feature time series per (recording, band) group:
// scripts/tda_eeg_audio_comparison.py:104-114 -> scipy.stats.spearmanr = Pearson correlation
// (np.corrcoef) of the average ranks.  x, y: (n_total, ld) rows = windows; column `col`.
// r = 0 when the group has < 5 windows or either series has np.std <= 1e-10 (cmp:110-114).
// One thread per (group, column); groups are tiny (<= 15 windows in the reference).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
spearman_kernel(const double* __restrict__ x, const double* __restrict__ y, int ld, const int* __restrict__ cols,
                int n_cols, const int* __restrict__ seg_off, int n_seg, double* __restrict__ r_out)
{
    const int seg = blockIdx.x;
    const int ci = threadIdx.x;
    if (seg >= n_seg || ci >= n_cols) return;
    const int col = cols[ci];
    const int s0 = seg_off[seg], n = seg_off[seg + 1] - s0;
    const double* xa = x + (size_t)s0 * ld + col;
    const double* ya = y + (size_t)s0 * ld + col;
    double r = 0.0;
    if (n >= 5) {
        auto fx = [=](int i) { return xa[(size_t)i * ld]; };
        auto fy = [=](int i) { return ya[(size_t)i * ld]; };
        const double mx = np_pairwise_fn(fx, 0, n) / n, my = np_pairwise_fn(fy, 0, n) / n;
        auto vx = [=](int i) { const double z = xa[(size_t)i * ld] - mx; return z * z; };
        auto vy = [=](int i) { const double z = ya[(size_t)i * ld] - my; return z * z; };
        const double sx = sqrt(np_pairwise_fn(vx, 0, n) / n), sy = sqrt(np_pairwise_fn(vy, 0, n) / n);
        if (sx > 1e-10 && sy > 1e-10) {
            // average ranks (scipy.stats.rankdata, method="average"), centred: sum of ranks = n(n+1)/2
            const double mr = 0.5 * (double)(n + 1);
            double sxx = 0.0, syy = 0.0, sxy = 0.0;
            for (int i = 0; i < n; ++i) {
                const double xi = xa[(size_t)i * ld], yi = ya[(size_t)i * ld];
                int lx = 0, ex = 0, ly = 0, ey = 0;
                for (int j = 0; j < n; ++j) {
                    const double xj = xa[(size_t)j * ld], yj = ya[(size_t)j * ld];
                    lx += xj < xi; ex += xj == xi; ly += yj < yi; ey += yj == yi;
                }
                const double rx = (double)lx + 0.5 * (double)(ex + 1) - mr;
                const double ry = (double)ly + 0.5 * (double)(ey + 1) - mr;
                sxx += rx * rx; syy += ry * ry; sxy += rx * ry;
            }
            r = (sxy / sqrt(sxx)) / sqrt(syy);
            if (r > 1.0) r = 1.0;
            if (r < -1.0) r = -1.0;
        }
    }
    r_out[(size_t)seg * n_cols + ci] = r;
}

tda_status launch_spearman(tda_ctx* ctx, const double* x, const double* y, int ld, const int* cols, int n_cols,
                           const int* seg_off, int n_seg, double* r_out, hipStream_t st)
{
    if (n_seg == 0 || n_cols == 0) return TDA_OK;
    if (n_cols > 64) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "at most 64 columns");
    hipLaunchKernelGGL(spearman_kernel, dim3(n_seg), dim3(64), 0, st, x, y, ld, cols, n_cols, seg_off, n_seg, r_out);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

// ---------------------------------------------------------------------------------
tda_status launch_tau(tda_ctx* ctx, const double* win, int n_win, int n_t, int max_lag, int* tau, hipStream_t st)
{
    if (n_win == 0) return TDA_OK;
    if (n_t < 2 || n_t > 8192) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "n_t must be in [2,8192]");
    hipLaunchKernelGGL(tau_kernel, dim3(n_win), dim3(64), (size_t)n_t * 8, st, win, n_win, n_t, max_lag, tau,
                       (const int*)nullptr, (int*)nullptr);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

tda_status launch_tau_segments(tda_ctx* ctx, const double* win, const int* seg_off, int n_seg, int n_t, int max_lag,
                               int* tau_seg, int* tau_win, hipStream_t st)
{
    if (n_seg == 0) return TDA_OK;
    if (n_t < 2 || n_t > 8192) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "n_t must be in [2,8192]");
    hipLaunchKernelGGL(tau_kernel, dim3(n_seg), dim3(64), (size_t)n_t * 8, st, win, n_seg, n_t, max_lag, tau_seg, seg_off,
                       tau_win);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

tda_status launch_diagram_finish(tda_ctx* ctx, const tda_diagram_set* sets, int n_sets, int n_dgm, hipStream_t st)
{
    if (n_dgm == 0 || n_sets == 0) return TDA_OK;
    if (n_sets < 0 || n_sets > 4) TDA_FAIL(ctx, TDA_ERR_INVALID, "1 to 4 diagram sets per launch");
    DiagramSets S;
    int cap = 1;
    for (int i = 0; i < 4; ++i) {
        const bool on = i < n_sets;
        S.rows[i] = on ? sets[i].rows : nullptr; S.cnt[i] = on ? sets[i].cnt : nullptr;
        S.cap[i] = on ? sets[i].cap : 0; S.order[i] = on ? sets[i].order : 0; S.feat[i] = on ? sets[i].feat : nullptr;
        if (on) {
            if (!sets[i].rows || !sets[i].cnt) TDA_FAIL(ctx, TDA_ERR_INVALID, "null diagram set");
            if (sets[i].cap < 1 || sets[i].cap > 4096) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "diagram capacity must be in [1,4096]");
            cap = sets[i].cap > cap ? sets[i].cap : cap;
        }
    }
    S.n_sets = n_sets;
    const long long n_all = (long long)n_sets * n_dgm;
    // small diagrams first (see the kernel), then whatever is larger with slices of the full capacity
    static const bool one_launch = getenv("TDA_FINISH_ONE_LAUNCH") != nullptr;
    int k_lo = INT_MIN;
    if (cap > 64 && !one_launch) {
        const int nw = FIN_WAVES;
        hipLaunchKernelGGL(diagram_finish_kernel, dim3((unsigned)((n_all + nw - 1) / nw)), dim3(64 * nw), (size_t)64 * 4 * 8 * nw, st, S,
                           n_dgm, 64, k_lo);
        k_lo = 64;
    }
    int nw = FIN_WAVES;                                  // diagrams per workgroup: fewer when the slices are large
    while (nw > 1 && (size_t)cap * 4 * 8 * nw > 32 * 1024) nw >>= 1;
    const size_t lds = (size_t)cap * 4 * 8 * nw;
    if (lds > 48 * 1024)
        TDA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(diagram_finish_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(diagram_finish_kernel, dim3((unsigned)((n_all + nw - 1) / nw)), dim3(64 * nw), lds, st, S, n_dgm, cap, k_lo);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

tda_status launch_features(tda_ctx* ctx, const double* dgm, const int* cnt, int n_dgm, int cap, double* feat,
                           hipStream_t st)
{
    if (cap < 1 || cap > 2048) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "diagram capacity must be in [1,2048]");
    tda_diagram_set one{const_cast<double*>(dgm), cnt, cap, 0, feat};
    return launch_diagram_finish(ctx, &one, 1, n_dgm, st);
}

tda_status launch_aggregate(tda_ctx* ctx, const double* f0, const double* f1, const int* seg_off, int n_seg,
                            double* out, hipStream_t st)
{
    if (n_seg == 0) return TDA_OK;
    hipLaunchKernelGGL(aggregate_kernel, dim3(n_seg), dim3(64), 0, st, f0, f1, seg_off, n_seg, out);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

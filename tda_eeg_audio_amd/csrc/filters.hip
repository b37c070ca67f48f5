// filters.hip -- zero-phase IIR filtering of many signals at once (SURVEY.md section 8f "next" rows 2-3).
//
//   sosfiltfilt_kernel  replaces scipy.signal.sosfiltfilt(sos, x) as called per channel by
//                       apply_bandpass_filter (notebooks/1_preprocesamiento.ipynb:236-263, sos from
//                       design_bandpass_filter nb1:209-233): the EEG band-pass in front of the windows.
//   filtfilt_kernel     replaces scipy.signal.filtfilt(b, a, s) of bandpass_filter
//                       (scripts/utils.py:66-74): the per-band filter of the audio envelope.
//
// Both follow scipy's algorithm exactly: odd extension by `edge` samples at both ends, forward
// pass started from zi * ext[0], backward pass over the reversed output started from zi * y[-1],
// trim.  The recursions are direct form II transposed with scipy's operation order (no fused
// multiply-add; the file is built with -ffp-contract=off), so results are bit-identical to scipy on
// the same float64 input.  Filter DESIGN (butter -> sos / ba, sosfilt_zi / lfilter_zi, edge) is
// host-side preparation done with scipy itself.
//
// The recursion is sequential in time, so parallelism comes from the signals: one lane per signal,
// 64 signals per workgroup.  Samples move through a 64 x 64 LDS tile so that HBM is read and written
// in rows (coalesced) while each lane walks its own row of the tile.
#include "common.h"
#include <stdlib.h>

#define FT 64            // tile: FT signals x FT samples
#define FTP (FT + 1)     // padded row (doubles): lane-per-row walks are bank-conflict free
#define F_MAX_SEC 8      // second-order sections
#define F_MAX_TAPS 17    // len(b) = len(a) of the ba form

struct SosParams {
    double c[F_MAX_SEC][6];   // b0 b1 b2 a0 a1 a2 per section
    double zi[F_MAX_SEC][2];
    int n_sec;
};
struct BaParams {
    double b[F_MAX_TAPS], a[F_MAX_TAPS];   // already divided by a[0]
    double zi[F_MAX_TAPS];                 // lfilter_zi, length ntaps-1
    int ntaps;
};
// A BANK of filters over the same signals (the five frequency bands of the path): blockIdx.y picks the filter, the
// outputs of filter f lie behind those of filter f - 1.  One launch instead of five -- the recursions are chains of
// dependent operations on few waves, and five times the waves cost (almost) no more time.
#define F_MAX_BANK 5
struct SosBank { SosParams f[F_MAX_BANK]; };
struct BaBank { BaParams f[F_MAX_BANK]; };

// odd extension (scipy.signal._arraytools.odd_ext): index i of the padded signal, N = L + 2*edge
__device__ __forceinline__ double odd_ext_at(const double* __restrict__ x, int L, int edge, int i)
{
    if (i < edge) return 2.0 * x[0] - x[edge - i];
    if (i < edge + L) return x[i - edge];
    const int k = i - (edge + L);
    return 2.0 * x[L - 1] - x[L - 2 - k];
}

// FILTER::step(state, x) -> y ; FILTER::init(state, params, x0)
struct SosFilter {
    double z[F_MAX_SEC][2];
    __device__ __forceinline__ void init(const SosParams& p, double x0)
    {
#pragma unroll
        for (int s = 0; s < F_MAX_SEC; ++s) { z[s][0] = p.zi[s][0] * x0; z[s][1] = p.zi[s][1] * x0; }
    }
    __device__ __forceinline__ double step(const SosParams& p, double x_cur)
    {
        // scipy/signal/_sosfilt.pyx: x_new = b0*x + z0; z0 = (b1*x - a1*x_new + z1); z1 = (b2*x - a2*x_new)
#pragma unroll
        for (int s = 0; s < F_MAX_SEC; ++s) {
            if (s >= p.n_sec) break;                       // uniform: state stays in registers
            const double x_new = p.c[s][0] * x_cur + z[s][0];
            z[s][0] = (p.c[s][1] * x_cur - p.c[s][4] * x_new) + z[s][1];
            z[s][1] = p.c[s][2] * x_cur - p.c[s][5] * x_new;
            x_cur = x_new;
        }
        return x_cur;
    }
};

// ND delays (ND = 8 for the reference's 4th-order band-pass and low-pass: ntaps = 9; 16 otherwise)
template <int ND>
struct BaFilterN {
    double z[ND];
    __device__ __forceinline__ void init(const BaParams& p, double x0)
    {
#pragma unroll
        for (int k = 0; k < ND; ++k) z[k] = p.zi[k] * x0;
    }
    __device__ __forceinline__ double step(const BaParams& p, double xn)
    {
        // scipy/signal/_lfilter.c.in (DOUBLE_filt): yn = Z[0] + b0*xn; Z[n] = Z[n+1] + xn*b[n+1] - yn*a[n+1]
        // b, a, zi are zero beyond ntaps, so running all ND >= ntaps - 1 delays is the same recursion:
        // the delay at ntaps-2 reads z[ntaps-1] = 0 and delays above stay 0 (their +0.0 never changes a sum
        // except the sign of an exact zero, which no later operation can observe in y)
        const double yn = z[0] + p.b[0] * xn;
#pragma unroll
        for (int n = 0; n < ND - 1; ++n) z[n] = (z[n + 1] + xn * p.b[n + 1]) - yn * p.a[n + 1];
        z[ND - 1] = xn * p.b[ND] - yn * p.a[ND];
        return yn;
    }
};
typedef BaFilterN<F_MAX_TAPS - 1> BaFilter;
typedef BaFilterN<8> BaFilter8;

template <class FILT, class BANK>
__global__ void __launch_bounds__(FT)
zero_phase_kernel(const double* __restrict__ x, int n_sig, int L, int edge, BANK bank, double* __restrict__ y,
                  double* __restrict__ work)
{
    __shared__ double tile[FT * FTP];
    const auto& p = bank.f[blockIdx.y];
    const int lane = threadIdx.x;
    const int s0 = blockIdx.x * FT;
    const int sig = s0 + lane;
    const int N = L + 2 * edge;
    y += (size_t)blockIdx.y * n_sig * L;
    work += (size_t)blockIdx.y * n_sig * N;
    const bool live = sig < n_sig;
    FILT f;
    // The kernel is a chain of dependent chunks on very few waves (one lane per signal): the rows of the NEXT chunk are
    // fetched into registers (64 per lane: rows s0..s0+63, sample c0 + lane of each -- coalesced along the rows) before
    // the lanes walk the current one, so that the load latency hides behind the recursion.
    double v[FT];
    auto fetch_fwd = [&](int c0) {
        const int cn = (N - c0) < FT ? (N - c0) : FT;
#pragma unroll
        for (int r = 0; r < FT; ++r)
            v[r] = (s0 + r < n_sig && lane < cn) ? odd_ext_at(x + (size_t)(s0 + r) * L, L, edge, c0 + lane) : 0.0;
    };
    auto fetch_bwd = [&](int c1) {
        const int c0 = c1 - FT > 0 ? c1 - FT : 0;
        const int cn = c1 - c0;
#pragma unroll
        for (int r = 0; r < FT; ++r)
            v[r] = (s0 + r < n_sig && lane < cn) ? work[(size_t)(s0 + r) * N + c0 + lane] : 0.0;
    };
    // ---- forward over the odd extension, output to work (n_sig, N) ----
    fetch_fwd(0);
    for (int c0 = 0; c0 < N; c0 += FT) {
        const int cn = (N - c0) < FT ? (N - c0) : FT;
#pragma unroll
        for (int r = 0; r < FT; ++r) tile[r * FTP + lane] = v[r];
        __syncthreads();
        if (c0 + FT < N) fetch_fwd(c0 + FT);
        if (live) {
            if (c0 == 0) f.init(p, tile[lane * FTP]);
            for (int t = 0; t < cn; ++t) tile[lane * FTP + t] = f.step(p, tile[lane * FTP + t]);
        }
        __syncthreads();
        for (int r = 0; r < FT; ++r)
            if (s0 + r < n_sig && lane < cn) work[(size_t)(s0 + r) * N + c0 + lane] = tile[r * FTP + lane];
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();                                    // this workgroup's rows of `work` are complete
    // ---- backward over work, trimmed result to y (n_sig, L) ----
    fetch_bwd(N);
    for (int c1 = N; c1 > 0; c1 -= FT) {
        const int c0 = c1 - FT > 0 ? c1 - FT : 0;
        const int cn = c1 - c0;
#pragma unroll
        for (int r = 0; r < FT; ++r) tile[r * FTP + lane] = v[r];
        __syncthreads();
        if (c0 > 0) fetch_bwd(c0);
        if (live) {
            if (c1 == N) f.init(p, tile[lane * FTP + cn - 1]);       // zi * y[-1]
            for (int t = cn - 1; t >= 0; --t) tile[lane * FTP + t] = f.step(p, tile[lane * FTP + t]);
        }
        __syncthreads();
        for (int r = 0; r < FT; ++r) {
            const int i = c0 + lane;                                  // index in the padded signal
            if (s0 + r < n_sig && lane < cn && i >= edge && i < edge + L) y[(size_t)(s0 + r) * L + i - edge] = tile[r * FTP + lane];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// The second-order-section cascade with its sections PIPELINED across lanes: LPS lanes per signal (LPS = the number
// of sections rounded up to a power of two), lane s runs section s one sample behind lane s-1 and takes its input
// from that lane's previous output over the DPP network (row_shr:1).  Every section executes scipy's recursion
// unchanged and in the same order for its own samples, so the output stays bit-identical to sosfiltfilt; what
// changes is the dependent chain per sample -- one section instead of the whole cascade -- and the number of
// waves: 64 / LPS signals per wave instead of 64 (the 16,638 channels of 354 recordings: 1,040 waves instead of
// 260, on 1,024 SIMDs).  Sections beyond n_sec are the identity (b0 = 1, everything else 0: x_new = 1 * x + 0).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dpp_row_shr1_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int LPS>
__global__ void __launch_bounds__(64, 4)     // (one wave per workgroup, 13 of them per CU for a shard of the corpus: 128 VGPRs)
sos_pipe_kernel(const double* __restrict__ x, int n_sig, int L, int edge, SosBank bank, double* __restrict__ y,
                double* __restrict__ work)
{
    constexpr int SPW = 64 / LPS;                      // signals per wave
    // ONE tile: the last section's lane writes output position t - (LPS - 1) over input position t in the step in
    // which the first section's lane has read it (a wave's LDS accesses are served in program order).  Two tiles were
    // 17 KB per one-wave workgroup: nine per CU where a shard of the corpus brings 13.5 -- a second, half-empty round
    __shared__ double tin[SPW * FTP];
    double* tout = tin;
    __shared__ double coef[F_MAX_SEC][8];              // b0 b1 b2 a0 a1 a2 zi0 zi1 per section (identity beyond n_sec)
    const SosParams& p = bank.f[blockIdx.y];
    const int lane = threadIdx.x, sl = lane / LPS, s = lane % LPS;
    const int s0 = blockIdx.x * SPW, sig = s0 + sl;
    const int N = L + 2 * edge;
    y += (size_t)blockIdx.y * n_sig * L;
    work += (size_t)blockIdx.y * n_sig * N;
    const bool live = sig < n_sig;
    if (lane < F_MAX_SEC * 8) {
        const int q = lane >> 3, k = lane & 7;
        double v = (k == 0) ? 1.0 : 0.0;
        if (q < p.n_sec) v = k < 6 ? p.c[q][k] : p.zi[q][k - 6];
        coef[q][k] = v;
    }
    __syncthreads();
    const double b0 = coef[s][0], b1 = coef[s][1], b2 = coef[s][2], a1 = coef[s][4], a2 = coef[s][5];
    const double zi0 = coef[s][6], zi1 = coef[s][7];
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: forward over the odd extension -> work (n_sig, N); pass 1: backward over work -> y (trimmed)
        double x0 = 0.0;
        if (live) x0 = pass == 0 ? odd_ext_at(x + (size_t)sig * L, L, edge, 0) : work[(size_t)sig * N + N - 1];
        double z0 = zi0 * x0, z1 = zi1 * x0;            // sosfilt_zi * first input sample, every section
        double outp = 0.0;                              // this lane's output of the previous step
        // input chunk: positions c0 .. c0+63 of the pass (lane = position: coalesced rows).  The rows of the NEXT chunk are
        // fetched into registers before the steps of the current one: the load latency hides behind the recursion
        double v[SPW];
        auto fetch = [&](int c0) {
            const int j = c0 + lane;
#pragma unroll
            for (int r = 0; r < SPW; ++r) {
                v[r] = 0.0;
                if (s0 + r < n_sig && j < N)
                    v[r] = pass == 0 ? odd_ext_at(x + (size_t)(s0 + r) * L, L, edge, j) : work[(size_t)(s0 + r) * N + (N - 1 - j)];
            }
        };
        fetch(0);
        for (int c0 = 0; c0 < N + LPS - 1; c0 += FT) {
#pragma unroll
            for (int r = 0; r < SPW; ++r) tin[r * FTP + lane] = v[r];
            __syncthreads();
            if (c0 + FT < N + LPS - 1) fetch(c0 + FT);
            double x_in = tin[sl * FTP];                // the first section's input, fetched one step ahead (the row is FT + 1 long)
            for (int t = 0; t < FT; ++t) {
                const int i = c0 + t - s;               // position this lane works on in this step
                const double from_prev = dpp_row_shr1_f64(outp);
                const double x_cur = s == 0 ? x_in : from_prev;
                x_in = tin[sl * FTP + t + 1];
                if (i >= 0 && i < N) {
                    // scipy/signal/_sosfilt.pyx: x_new = b0*x + z0; z0 = (b1*x - a1*x_new + z1); z1 = (b2*x - a2*x_new)
                    const double x_new = b0 * x_cur + z0;
                    z0 = (b1 * x_cur - a1 * x_new) + z1;
                    z1 = b2 * x_cur - a2 * x_new;
                    outp = x_new;
                }
                if (s == LPS - 1) tout[sl * FTP + t] = outp;        // position c0 + t - (LPS - 1) of the output
            }
            __syncthreads();
            for (int r = 0; r < SPW; ++r) {
                const int j = c0 + lane - (LPS - 1);                // output position of the pass
                if (s0 + r < n_sig && j >= 0 && j < N) {
                    const double v = tout[r * FTP + lane];
                    if (pass == 0) work[(size_t)(s0 + r) * N + j] = v;
                    else {
                        const int i = N - 1 - j;                    // index in the padded signal
                        if (i >= edge && i < edge + L) y[(size_t)(s0 + r) * L + i - edge] = v;
                    }
                }
            }
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();                                // work complete (this workgroup's rows) before the backward pass
    }
}

tda_status launch_sosfiltfilt(tda_ctx* ctx, const double* x, int n_sig, int L, const double* sos, const double* zi,
                              int n_sec, int edge, double* y, double* work, hipStream_t st, int n_filt)
{
    if (n_sig == 0 || n_filt == 0) return TDA_OK;
    if (n_sec < 1 || n_sec > F_MAX_SEC) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "n_sections must be in [1,8]");
    if (n_filt < 1 || n_filt > F_MAX_BANK) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "a filter bank holds 1..5 filters");
    if (edge < 0 || L <= edge) TDA_FAIL(ctx, TDA_ERR_INVALID, "signal length must exceed the pad length");   // scipy raises too
    SosBank bank = {};
    for (int f = 0; f < n_filt; ++f) {
        SosParams& p = bank.f[f];
        p.n_sec = n_sec;
        for (int s = 0; s < n_sec; ++s) {
            for (int k = 0; k < 6; ++k) p.c[s][k] = sos[(f * n_sec + s) * 6 + k];
            p.zi[s][0] = zi[(f * n_sec + s) * 2]; p.zi[s][1] = zi[(f * n_sec + s) * 2 + 1];
        }
    }
    // sections pipelined across lanes (bit-identical; see sos_pipe_kernel).  TDA_SOS_SERIAL=1: the one-lane-per-signal
    // form, kept for measurements
    static const bool serial = getenv("TDA_SOS_SERIAL") != nullptr;
    if (serial)
        hipLaunchKernelGGL((zero_phase_kernel<SosFilter, SosBank>), dim3((n_sig + FT - 1) / FT, n_filt), dim3(FT), 0, st, x, n_sig,
                           L, edge, bank, y, work);
    else if (n_sec <= 2)
        hipLaunchKernelGGL(sos_pipe_kernel<2>, dim3((n_sig + 31) / 32, n_filt), dim3(64), 0, st, x, n_sig, L, edge, bank, y, work);
    else if (n_sec <= 4)
        hipLaunchKernelGGL(sos_pipe_kernel<4>, dim3((n_sig + 15) / 16, n_filt), dim3(64), 0, st, x, n_sig, L, edge, bank, y, work);
    else
        hipLaunchKernelGGL(sos_pipe_kernel<8>, dim3((n_sig + 7) / 8, n_filt), dim3(64), 0, st, x, n_sig, L, edge, bank, y, work);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

// ---------------------------------------------------------------------------------------------------
// The (b, a) recursion with its DELAYS spread over lanes: 8 lanes per signal, lane k keeps z[k].  A step of scipy's
// direct form II transposed is  yn = z[0] + b0 xn;  z[k] = (z[k+1] + xn b[k+1]) - yn a[k+1]  -- the eight updates are
// independent of each other once yn is known, so every lane does ONE of them: z[0] is broadcast inside the group of
// eight (two DPP moves per half), every lane forms the same yn from it, z[k+1] comes from the neighbour lane (row_shl:1).
// Each delay sees exactly the operations of BaFilterN<8>::step in the same order: bit-identical to scipy.signal.lfilter
// / filtfilt.  One lane per signal (zero_phase_kernel<BaFilter8>) is a chain of ~30 instructions per sample on 19 waves
// for the 5 x 236 envelopes of a shard of recordings -- 5 ms with the chip idle, the longest item of the raw-recordings
// leg; here it is ~15 per sample on 150 waves.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dpp_bcast8_f64(double v)
{
    // lane & ~7 of every group of eight: quad_perm [0,0,0,0], then the odd quads take the quad before them (row_shr:4)
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x00, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x00, 0xF, 0xF, false);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x114, 0xF, 0xA, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x114, 0xF, 0xA, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_row_shl1_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x101, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x101, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(64)
ba_pipe_kernel(const double* __restrict__ x, int n_sig, int L, int edge, BaBank bank, double* __restrict__ y,
               double* __restrict__ work)
{
    constexpr int ND = 8, SPW = 64 / ND;               // delays = lanes per signal; signals per wave
    __shared__ double tile[SPW * FTP];
    const BaParams& p = bank.f[blockIdx.y];
    const int lane = threadIdx.x, g = lane / ND, k = lane % ND;
    const int s0 = blockIdx.x * SPW;
    const int N = L + 2 * edge;
    y += (size_t)blockIdx.y * n_sig * L;
    work += (size_t)blockIdx.y * n_sig * N;
    const double b0 = p.b[0], bk = p.b[k + 1], ak = p.a[k + 1], zik = p.zi[k];
    const bool last = k == ND - 1;
    double* row = tile + g * FTP;
    double z = 0.0;
    auto step = [&](int t) {
        const double xn = row[t];                       // (one address per group: an LDS broadcast)
        const double yn = dpp_bcast8_f64(z) + b0 * xn;
        const double t1 = xn * bk;
        const double zn = dpp_row_shl1_f64(z);          // z[k + 1]; the last delay has none (scipy: x b[last] - y a[last])
        const double u = last ? t1 : zn + t1;
        z = u - yn * ak;
        if (k == 0) row[t] = yn;
    };
    double v[SPW];
    auto fetch_fwd = [&](int c0) {
        const int cn = (N - c0) < FT ? (N - c0) : FT;
#pragma unroll
        for (int r = 0; r < SPW; ++r)
            v[r] = (s0 + r < n_sig && lane < cn) ? odd_ext_at(x + (size_t)(s0 + r) * L, L, edge, c0 + lane) : 0.0;
    };
    auto fetch_bwd = [&](int c1) {
        const int c0 = c1 - FT > 0 ? c1 - FT : 0;
        const int cn = c1 - c0;
#pragma unroll
        for (int r = 0; r < SPW; ++r)
            v[r] = (s0 + r < n_sig && lane < cn) ? work[(size_t)(s0 + r) * N + c0 + lane] : 0.0;
    };
    // ---- forward over the odd extension, output to work (n_sig, N) ----
    fetch_fwd(0);
    for (int c0 = 0; c0 < N; c0 += FT) {
        const int cn = (N - c0) < FT ? (N - c0) : FT;
#pragma unroll
        for (int r = 0; r < SPW; ++r) tile[r * FTP + lane] = v[r];
        __syncthreads();
        if (c0 + FT < N) fetch_fwd(c0 + FT);
        if (c0 == 0) z = zik * row[0];                  // lfilter_zi * x[0]
        for (int t = 0; t < cn; ++t) step(t);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SPW; ++r)
            if (s0 + r < n_sig && lane < cn) work[(size_t)(s0 + r) * N + c0 + lane] = tile[r * FTP + lane];
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();                                    // this workgroup's rows of `work` are complete
    // ---- backward over work, trimmed result to y (n_sig, L) ----
    fetch_bwd(N);
    for (int c1 = N; c1 > 0; c1 -= FT) {
        const int c0 = c1 - FT > 0 ? c1 - FT : 0;
        const int cn = c1 - c0;
#pragma unroll
        for (int r = 0; r < SPW; ++r) tile[r * FTP + lane] = v[r];
        __syncthreads();
        if (c0 > 0) fetch_bwd(c0);
        if (c1 == N) z = zik * row[cn - 1];             // zi * y[-1]
        for (int t = cn - 1; t >= 0; --t) step(t);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SPW; ++r) {
            const int i = c0 + lane;                    // index in the padded signal
            if (s0 + r < n_sig && lane < cn && i >= edge && i < edge + L) y[(size_t)(s0 + r) * L + i - edge] = tile[r * FTP + lane];
        }
        __syncthreads();
    }
}

tda_status launch_filtfilt(tda_ctx* ctx, const double* x, int n_sig, int L, const double* b, const double* a,
                           const double* zi, int ntaps, int edge, double* y, double* work, hipStream_t st, int n_filt)
{
    if (n_sig == 0 || n_filt == 0) return TDA_OK;
    if (ntaps < 2 || ntaps > F_MAX_TAPS) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "len(b)=len(a) must be in [2,17]");
    if (n_filt < 1 || n_filt > F_MAX_BANK) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "a filter bank holds 1..5 filters");
    if (edge < 0 || L <= edge) TDA_FAIL(ctx, TDA_ERR_INVALID, "signal length must exceed the pad length");
    BaBank bank = {};
    for (int f = 0; f < n_filt; ++f) {
        BaParams& p = bank.f[f];
        p.ntaps = ntaps;
        const double a0 = a[f * ntaps];
        for (int k = 0; k < F_MAX_TAPS; ++k) { p.b[k] = k < ntaps ? b[f * ntaps + k] / a0 : 0.0; p.a[k] = k < ntaps ? a[f * ntaps + k] / a0 : 0.0; }
        for (int k = 0; k < F_MAX_TAPS; ++k) p.zi[k] = k < ntaps - 1 ? zi[f * (ntaps - 1) + k] : 0.0;
    }
    // ntaps <= 9 (the reference's 4th-order designs): the delays spread over lanes (bit-identical; see ba_pipe_kernel).
    // TDA_BA_SERIAL=1: the one-lane-per-signal form, kept for measurements
    static const bool serial = getenv("TDA_BA_SERIAL") != nullptr;
    if (ntaps <= 9 && !serial)
        hipLaunchKernelGGL(ba_pipe_kernel, dim3((n_sig + 7) / 8, n_filt), dim3(64), 0, st, x, n_sig, L, edge, bank, y, work);
    else if (ntaps <= 9)
        hipLaunchKernelGGL((zero_phase_kernel<BaFilter8, BaBank>), dim3((n_sig + FT - 1) / FT, n_filt), dim3(FT), 0, st, x, n_sig,
                           L, edge, bank, y, work);
    else
        hipLaunchKernelGGL((zero_phase_kernel<BaFilter, BaBank>), dim3((n_sig + FT - 1) / FT, n_filt), dim3(FT), 0, st, x, n_sig,
                           L, edge, bank, y, work);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

// ---------------------------------------------------------------------------------------------------
// Audio front end (scripts/utils.py:56-79): rational resampling and Hilbert envelope.
//
//   upfirdn_kernel      replaces scipy.signal.resample_poly(audio, 250, 44100) (utils.py:77-79): the FIR
//                       low-pass h (Kaiser window, 2*10*max(up,down)+1 taps, padded as scipy pads it) is
//                       designed on the host with scipy; the kernel evaluates the polyphase sums
//                         y[j] = sum_i x[i] * h[(j + n_pre_remove)*down - i*up]
//                       one output sample per thread, i ascending.
//   hilbert_env_kernel  replaces np.abs(scipy.signal.hilbert(s)) (utils.py:58-59): the analytic signal is
//                       ifft(fft(x)*h) = x (*) ifft(h) (circular convolution); its real part is x and its
//                       imaginary part is x (*) g with g = imag(ifft(h)) tabulated on the host, so
//                         env[n] = sqrt(x[n]^2 + (sum_m x[m] g[(n-m) mod N])^2)
//                       -- an O(N^2) sum (N ~ 5,000 at 250 Hz), exact to rounding, no FFT needed.
// Both are float64 sums in a fixed order; agreement with scipy is to rounding (1e-12 relative in the
// tests), not bit-identical (scipy goes through pocketfft / a different accumulation order).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
upfirdn_kernel(const double* __restrict__ x, long long n_in, const double* __restrict__ h, int len_h, int up, int down,
               long long n_pre_remove, long long n_out, double* __restrict__ y)
{
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_out) return;
    const long long t = (j + n_pre_remove) * down;          // position in the up-sampled domain
    // taps k = t - i*up in [0, len_h)  <=>  i in [ceil((t-len_h+1)/up), floor(t/up)]
    long long i_hi = t / up;
    long long i_lo = t - (len_h - 1);
    i_lo = i_lo <= 0 ? 0 : (i_lo + up - 1) / up;
    if (i_hi > n_in - 1) i_hi = n_in - 1;
    double acc = 0.0;
    for (long long i = i_lo; i <= i_hi; ++i) acc += x[i] * h[t - i * up];
    y[j] = acc;
}

__global__ void __launch_bounds__(256)
hilbert_env_kernel(const double* __restrict__ x, int n, const double* __restrict__ g, double* __restrict__ env)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* xs = reinterpret_cast<double*>(smem);          // x tile
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    for (int m0 = 0; m0 < n; m0 += 256) {
        const int mm = m0 + threadIdx.x;
        xs[threadIdx.x] = mm < n ? x[mm] : 0.0;
        __syncthreads();
        if (i < n) {
            const int lim = (n - m0) < 256 ? (n - m0) : 256;
            int d = i - m0;                                 // (i - m) mod n for m = m0
            if (d < 0) d += n;
            for (int q = 0; q < lim; ++q) {
                acc += xs[q] * g[d];
                d = d == 0 ? n - 1 : d - 1;
            }
        }
        __syncthreads();
    }
    if (i < n) { const double re = x[i]; env[i] = sqrt(re * re + acc * acc); }
}

tda_status launch_upfirdn(tda_ctx* ctx, const double* x, long long n_in, const double* h, int len_h, int up, int down,
                          long long n_pre_remove, long long n_out, double* y, hipStream_t st)
{
    if (n_out <= 0) return TDA_OK;
    if (up < 1 || down < 1 || len_h < 1 || n_in < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "bad resampling geometry");
    hipLaunchKernelGGL(upfirdn_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, x, n_in, h, len_h, up, down,
                       n_pre_remove, n_out, y);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

tda_status launch_hilbert_env(tda_ctx* ctx, const double* x, int n, const double* g, double* env, hipStream_t st)
{
    if (n <= 0) return TDA_OK;
    hipLaunchKernelGGL(hilbert_env_kernel, dim3((n + 255) / 256), dim3(256), 256 * sizeof(double), st, x, n, g, env);
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

// corr_dist_dev.h -- the window -> covariance part of corr_dist_kernel as a device function, shared by
// corr_dist_kernel (corr_dist.hip: matrices to HBM) and eeg_window_kernel (rips.hip: the matrix never leaves LDS).
// See corr_dist.hip for the description of the two passes.
#pragma once
#include "common.h"

// 32 samples per tile since round 3 (64 before): the tile is the largest item of the fused EEG kernel's LDS (25 KB of
// 38) and twelve prefetch registers per lane; at 32 the kernel needs 26.9 KB and 80 VGPRs -- six workgroups per CU
// instead of four, fused kernel 4.15 -> 3.61 ms per 55,224 windows.  The sums run over the samples in the same order
// whatever the tile length (the accumulators wait in LDS between tiles): bit-identical results.
#ifndef CD_TCH
#define CD_TCH 32         // time samples per LDS tile
#define CD_TSH 5          // log2(CD_TCH)
#endif
#define CD_MAXCH 64

typedef double d4 __attribute__((ext_vector_type(4)));

// COMPILER FAULT (ROCm 7.2.0 hipcc, AMD clang 22, gfx950), proven in round 3 -- why no kernel that calls this may be
// built with its MFMA accumulators in AGPRs (launch bound of ONE wave per SIMD).  The result of
// v_mfma_f64_16x16x4_f64 may be read 18 wait states after issue at the earliest, and on gfx950 that is the software's
// business (no interlock).  With the accumulators in AGPRs the hazard recogniser emits, on the path where a wave has
// no second block in a tile (the conditional q = 1 of mfma_tile),
//       v_mfma_f64_16x16x4_f64 a[0:7], ...; s_and_b64; s_cbranch_vccnz; s_waitcnt lgkmcnt(0);
//       s_nop 0; v_accvgpr_read_b32 v19, a7; s_nop 11; v_accvgpr_read_b32 v18, a6; ...
// i.e. a7 -- the high half of accumulator element 3 -- is read FOUR states after the MFMA and keeps the value of one
// k-step earlier.  Measured (tools/probes/wide_waves_repro.py on a -DTDA_EEG_WIDE_WAVES=1 build): every distance that
// involves a channel with (ch & 15) >> 2 == 3 is off by up to 4e-2, everything else is exact; with `s_nop 15; s_nop 3`
// patched into the assembly in front of that one read (tools/probes/patch_mfma_hazard.sh) the same build is bit-exact.
// Only the streaming (RES = false) widening variants show the pattern (3 places in rips.hip's code object, none with
// VGPR accumulators); a wait spelled out in the source does not help, because the compiler places the AGPR copies in
// front of it.  tests/test_capi_and_host.py::test_mfma_kernels_keep_accumulators_in_vgprs pins NumAgprs == 0.


// global -> registers (issued early, consumed after the current tile has been used).  Element
// idx = tid + 256 k of a tile is (channel idx / CD_TCH, sample idx % CD_TCH): a wave reads 64 / CD_TCH rows of
// 8 CD_TCH contiguous bytes.
template <int NPRE>
__device__ __forceinline__ void cd_fetch(double (&pre)[NPRE], const double* __restrict__ X, int ld, int n_ch, int c0,
                                         int tc, int tid)
{
    // element k of this thread is (channel (tid >> CD_TSH) + CPK k, sample tid & (CD_TCH - 1)), CPK = 256 / CD_TCH
    // channels per round of the workgroup: ONE per-thread 32-bit offset and a uniform row base per k (kept as scalars)
    // -- written the obvious way, X[(size_t)ch * ld + c0 + t], the compiler keeps twelve 64-bit per-thread offsets
    // alive across the tile loops and spills them
    constexpr int CPK = 256 / CD_TCH;
    const int t = tid & (CD_TCH - 1), ch0 = tid >> CD_TSH;
    const int toff = ch0 * ld + t;
    const bool live = t < tc;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const double* rowk = X + (size_t)(CPK * k) * (size_t)ld + c0;     // uniform
        pre[k] = (live && ch0 + CPK * k < n_ch) ? rowk[toff] : 0.0;
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

#define CD_RES_CHUNKS (256 / CD_TCH)   // windows of up to CD_RES_CHUNKS * CD_TCH samples stay in registers between the passes


template <int NB>
struct CdLayout {
    static constexpr int CP = 16 * NB;                  // padded channel count
    static constexpr int CSP = CP + 1;                  // tile row stride
    static constexpr int NBLK = NB * (NB + 1) / 2;      // upper-triangular 16x16 blocks
    static constexpr int TILE = CD_TCH * CSP;           // doubles
    static constexpr int ACCS = NBLK * 256;             // doubles
    static constexpr size_t BYTES = sizeof(double) * ((size_t)TILE + ACCS + 2 * CP);
    // element (i, j), i <= j or not, of the product matrix in the accumulator image of the MFMA blocks
    __device__ static __forceinline__ int acc_index(int i, int j)
    {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int r = lo >> 4, s = hi >> 4;
        const int b = r * NB - ((r * (r - 1)) >> 1) + (s - r);
        return (b * 4 + ((lo & 15) >> 2)) * 64 + ((lo & 3) << 4) + (hi & 15);
    }
};

// Passes A and B for the window at X (n_ch rows of n_t samples, row stride ld), by a 256-thread workgroup.
// On return (after a barrier) LDS holds: accs = the centred products X X^T (MFMA block images, see acc_index),
// mean[], and sdev[i] = sqrt(cov(i,i)); the tile area [0, TILE) is free again.
template <int NB, bool RES>
__device__ __forceinline__ void cd_window_products(unsigned char* smem, const double* __restrict__ X, int n_ch, int n_t,
                                                   int ld)
{
    constexpr int CP = 16 * NB;                  // padded channel count
    constexpr int CSP = CP + 1;                  // tile row stride
    constexpr int NPRE = CP * CD_TCH / 256;      // tile elements per thread
    constexpr int NBLK = NB * (NB + 1) / 2;      // upper-triangular 16x16 blocks
    constexpr int MAXQ = (NBLK + 3) / 4;         // blocks per wave and tile
    constexpr int NBUF = RES ? CD_RES_CHUNKS : 1;
    double* tile = reinterpret_cast<double*>(smem);            // CD_TCH * CSP
    double* accs = tile + CD_TCH * CSP;                        // NBLK * 256: accumulators between tiles
    double* mean = accs + NBLK * 256;                          // CP
    double* sdev = mean + CP;                                  // CP
    const int tid = threadIdx.x, l = tid & 63, wv = uni(tid >> 6);
    // window w = n_ch rows of n_t samples, row stride ld, starting win_stride elements after window w-1:
    //   stacked windows (preprocessed/<band>.npy): win_stride = n_ch*n_t, ld = n_t
    //   sliding windows over one band-passed recording (n_ch, L): win_stride = step, ld = L -- the 75 %
    //   overlap (nb1:338-341) is then served by L2 instead of being materialised 4x in HBM
    const int n_chunks = (n_t + CD_TCH - 1) / CD_TCH;
    double pre[NBUF][NPRE];
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
                cd_stash<NPRE, CSP>(pre[c], tile, mean, n_ch, chunk_len(c), tid, false);
                __syncthreads();
                sum_tile(chunk_len(c));
                __syncthreads();
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
                cd_stash<NPRE, CSP>(pre[c], tile, mean, n_ch, chunk_len(c), tid, true);
                __syncthreads();
                mfma_tile(c, chunk_len(c));
                __syncthreads();
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

    const double fact = 1.0 / (double)(n_t - 1);
    if (tid < n_ch) sdev[tid] = sqrt(accs[CdLayout<NB>::acc_index(tid, tid)] * fact);
    __syncthreads();
}

// wasserstein.hip -- batched order-1 Wasserstein distance between persistence diagrams.
//
// Replaces safe_wasserstein (scripts/utils.py:180-191) -> persim.wasserstein(dgm1, dgm2):
// Euclidean ground metric (sklearn's |x|^2 - 2x.y + |y|^2 expansion, float64), every point
// may instead go to the diagonal at cost (d-b)cos(pi/4)-ish (persim's 45-degree rotation),
// result = plain sum of matched costs.
//
// persim solves one (M+N)x(M+N) assignment with +inf blocks.  The same optimum is the
// min over PARTIAL matchings mu of  sum_mu C_ij + sum_{i not in mu} s_i + sum_{j not in mu} t_j
//   =  sum s + sum t + min_mu sum_mu (C_ij - s_i - t_j),
// i.e. a rectangular assignment (rows = the smaller diagram) with costs
// g_ij = min(0, C_ij - s_i - t_j) <= 0 and no forbidden entries.  It is solved EXACTLY in
// float64 by shortest augmenting paths with dual variables (Jonker-Volgenant class), one
// pair per wavefront: columns live on lanes (CW per lane), one Dijkstra step = CW LDS reads
// per lane + one wave min-reduction.  At most R(R+1)/2 steps for R rows.
// The returned value re-sums the ORIGINAL costs (C_ij, s_i, t_j) of the optimal matching.
#include "common.h"
#include <type_traits>

#ifdef TDA_PROFILE
__device__ unsigned long long g_prof_ws[16];
extern "C" __attribute__((visibility("default"))) int tda_profile_read_ws(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_ws), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof_ws), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#define WPROF(i, v) do { if (lane_id() == 0) atomicAdd(&g_prof_ws[i], (unsigned long long)(v)); } while (0)
#define WCLK() clock64()
#else
#define WPROF(i, v) do {} while (0)
#define WCLK() 0ull
#endif

#define WS_DEFERRED 0x40000000      // status of a pair between the small and the wide launch (never seen by the caller)
#define WS_CP 0.7071067811865476   // np.cos(np.pi/4)
#define WS_SP 0.7071067811865475   // np.sin(np.pi/4)

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_min_f64_ws(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// C(a_i, b_j) with a from the FIRST diagram and b from the SECOND, independent of which one
// plays the row role (mirrors oracle/tda_oracle.c::orc_wasserstein)
__device__ __forceinline__ double ws_cost(double ab, double ad, double bb, double bd)
{
    const double xx = ab * ab + ad * ad;
    const double yy = bb * bb + bd * bd;
    const double dot = fma(ad, bd, ab * bb);
    double d2 = -2.0 * dot;
    d2 += xx;
    d2 += yy;
    if (!(d2 > 0.0)) d2 = 0.0;
    return sqrt_rn(d2);                         // (sqrt() bit for bit, six instructions less: common.h)
}

template <int CW>
__global__ void __launch_bounds__(64)
wasserstein_kernel(const double* __restrict__ dgm_a, const int* __restrict__ cnt_a, int cap_a,
                   const double* __restrict__ dgm_b, const int* __restrict__ cnt_b, int cap_b,
                   const int* __restrict__ idx_a, const int* __restrict__ idx_b, int n_pairs,
                   int max_rows, int max_cols,
                   double* __restrict__ out, int* __restrict__ status, int mode)
{
    // mode 1: the SMALL first launch (LDS for 64 x 64 points whatever the capacities of the diagram buffers: four times
    // the workgroups per CU of a launch sized by a capacity of 256) leaves pairs that do not fit marked WS_DEFERRED;
    // mode 2: the launch sized by the capacities takes exactly those; mode 0: one launch for everything
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int pr = blockIdx.x;
    if (pr >= n_pairs) return;
    if (mode == 2 && status[pr] != WS_DEFERRED) return;
    const int lane = lane_id();
    unsigned long long wt0 = WCLK();
    (void)wt0;
    // LDS: row points (b,d,s) | u[rows] | col points (b,d,t) | owner[cols] | |y|^2 of the columns (1-D path)
    double* rb = reinterpret_cast<double*>(smem);
    double* rd = rb + max_rows;
    double* rs = rd + max_rows;
    double* ru = rs + max_rows;
    double* cb = ru + max_rows;
    double* cd = cb + max_cols;
    double* ct = cd + max_cols;
    int* owner = reinterpret_cast<int*>(ct + max_cols);                 // max_cols ints (padded to 8 B)
    double* G = ct + max_cols + ((max_cols + 1) >> 1);                  // max_cols doubles

    const int ia = idx_a ? idx_a[pr] : pr;
    const int ib = idx_b ? idx_b[pr] : pr;
    const double* A = dgm_a + (size_t)ia * cap_a * 2;
    const double* B = dgm_b + (size_t)ib * cap_b * 2;
    int ka = cnt_a[ia]; ka = ka < cap_a ? ka : cap_a; ka = ka < 0 ? 0 : ka;
    int kb = cnt_b[ib]; kb = kb < cap_b ? kb : cap_b; kb = kb < 0 ? 0 : kb;
    // count finite rows (utils.py:185-186)
    int M = 0, N = 0;
    for (int i0 = 0; i0 < ka; i0 += 64) {
        const int i = i0 + lane;
        M += __popcll(__ballot(i < ka && isfinite(A[2 * i]) && isfinite(A[2 * i + 1])));
    }
    for (int i0 = 0; i0 < kb; i0 += 64) {
        const int i = i0 + lane;
        N += __popcll(__ballot(i < kb && isfinite(B[2 * i]) && isfinite(B[2 * i + 1])));
    }
    const int Me = M > 0 ? M : 1, Ne = N > 0 ? N : 1;    // empty -> {(0,0)}  (utils.py:184,187)
    // rows = smaller diagram
    const bool a_is_row = Me <= Ne;
    const int R = a_is_row ? Me : Ne, Cn = a_is_row ? Ne : Me;
    const int cw_used = (Cn + 63) >> 6;                // column slots per lane actually in use
    if (R > max_rows || Cn > max_cols || Cn > 64 * CW) {
        if (lane == 0) {
            if (mode == 1) status[pr] = WS_DEFERRED;
            else { out[pr] = __longlong_as_double(0x7ff8000000000000ll); status[pr] = TDA_WIN_NOT_CONVERGED; }
        }
        return;
    }
    // load finite points, preserving order
    {
        double* pb = a_is_row ? rb : cb; double* pd = a_is_row ? rd : cd; double* pc = a_is_row ? rs : ct;
        int m = 0;
        for (int i0 = 0; i0 < ka; i0 += 64) {
            const int i = i0 + lane;
            double b = 0, d = 0; bool fin = false;
            if (i < ka) { b = A[2 * i]; d = A[2 * i + 1]; fin = isfinite(b) && isfinite(d); }
            const u64 bal = __ballot(fin);
            const int pos = m + __popcll(bal & ((1ull << lane) - 1ull));
            if (fin) { pb[pos] = b; pd[pos] = d; pc[pos] = fma(d, WS_CP, -(b * WS_SP)); }
            m += __popcll(bal);
        }
        if (M == 0 && lane == 0) { pb[0] = 0.0; pd[0] = 0.0; pc[0] = 0.0; }
    }
    {
        double* pb = a_is_row ? cb : rb; double* pd = a_is_row ? cd : rd; double* pc = a_is_row ? ct : rs;
        int m = 0;
        for (int i0 = 0; i0 < kb; i0 += 64) {
            const int i = i0 + lane;
            double b = 0, d = 0; bool fin = false;
            if (i < kb) { b = B[2 * i]; d = B[2 * i + 1]; fin = isfinite(b) && isfinite(d); }
            const u64 bal = __ballot(fin);
            const int pos = m + __popcll(bal & ((1ull << lane) - 1ull));
            if (fin) { pb[pos] = b; pd[pos] = d; pc[pos] = fma(d, WS_CP, -(b * WS_SP)); }
            m += __popcll(bal);
        }
        if (N == 0 && lane == 0) { pb[0] = 0.0; pd[0] = 0.0; pc[0] = 0.0; }
    }
    __syncthreads();

    auto cfull = [&](int i, int j) -> double {   // original point-to-point cost of row i, col j
        return a_is_row ? ws_cost(rb[i], rd[i], cb[j], cd[j]) : ws_cost(cb[j], cd[j], rb[i], rd[i]);
    };
    auto gain = [&](int i, int j) -> double {
        const double g = cfull(i, j) - rs[i] - ct[j];
        return g < 0.0 ? g : 0.0;
    };
    for (int i = lane; i < R; i += 64) ru[i] = 0.0;
    __syncthreads();
    WPROF(0, WCLK() - wt0); wt0 = WCLK();
    WPROF(4, 1); WPROF(5, R); WPROF(6, Cn);

    // ---- 1-D fast path ------------------------------------------------------------------
    // If every point of both diagrams has the same birth (two H0 diagrams: births 0) the points lie
    // on a line, the ground cost is |d_i - d_j| and, for any fixed sets of matched points, the sorted
    // (non-crossing) pairing is optimal.  The optimum over partial matchings is then the edit-distance
    // style recurrence  F[i][j] = min(F[i-1][j], F[i][j-1], F[i-1][j-1] + g_ij)  over death-sorted
    // diagrams - an anti-diagonal wavefront with one row per lane, R + C - 1 steps instead of
    // hundreds of Dijkstra steps.  Same gains g_ij as the general solver (so the same rounding of
    // C_ij); the value differs from the assignment optimum by at most the rounding noise of C.
    {
        bool ok = R <= 64;
        const double b0 = rb[0];
        for (int i0 = 0; i0 < R; i0 += 64) {
            const int i = i0 + lane;
            ok = ok && !__ballot(i < R && (rb[i] != b0 || (i + 1 < R && rd[i + 1] < rd[i])));
        }
        for (int j0 = 0; j0 < Cn; j0 += 64) {
            const int j = j0 + lane;
            ok = ok && !__ballot(j < Cn && (cb[j] != b0 || (j + 1 < Cn && cd[j + 1] < cd[j])));
        }
        if (ok) {
            double cur = 0.0, nb1 = 0.0, nb2 = 0.0;      // own F, neighbour row's F one / two steps ago
            const int nsteps1d = R + Cn - 1;
            // gains on the fly: the row's part of sklearn's expansion is a per-lane constant, the column's |y|^2 is
            // tabulated once -- the same operations in the same order as ws_cost(), 40 instead of 58 instructions per cell
            for (int j = lane; j < Cn; j += 64) G[j] = cb[j] * cb[j] + cd[j] * cd[j];
            __syncthreads();
            const int li = lane < R ? lane : 0;
            const double r_b = rb[li], r_d = rd[li], r_s = rs[li], r_n = r_b * r_b + r_d * r_d;
            // every birth equals b0: r_b * c_b is one constant; which diagram comes first in the sums is uniform per pair,
            // so the wavefront exists twice instead of selecting per cell (34 -> 29 instructions per cell)
            const double bb = b0 * b0;
            auto wavefront = [&](auto AROW) {
                constexpr bool A_ROW = decltype(AROW)::value;
                for (int t = 0; t < nsteps1d; ++t) {
                    const int j0 = t - lane;
                    // neighbour (row lane-1) value of the previous step; row 0 sees the zero boundary
                    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(cur), 0x138, 0xF, 0xF, false);
                    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(cur), 0x138, 0xF, 0xF, false);
                    nb2 = nb1;
                    nb1 = __hiloint2double(hi, lo);
                    if (lane < R && j0 >= 0 && j0 < Cn) {
                        const double c_d = cd[j0], c_n = G[j0];
                        const double dot = fma(r_d, c_d, bb);                   // = fma(ad, bd, ab * bb) either way round
                        double d2 = -2.0 * dot;
                        d2 += A_ROW ? r_n : c_n;                                // |x|^2 of the FIRST diagram's point, then the second's
                        d2 += A_ROW ? c_n : r_n;
                        if (!(d2 > 0.0)) d2 = 0.0;
                        double g = sqrt_rn(d2) - r_s - ct[j0];
                        g = g < 0.0 ? g : 0.0;
                        const double up = nb1;                          // F[row][j0+1] of the row above
                        const double dg = (j0 == 0 ? 0.0 : nb2) + g;    // F[row above][j0] + g
                        double m = up < cur ? up : cur;                 // cur still holds F[row+1][j0] (left)
                        m = dg < m ? dg : m;
                        cur = m;
                    }
                }
            };
            if (a_is_row) wavefront(std::true_type{}); else wavefront(std::false_type{});
            const double fbest = uni_f64(cur, R - 1);
            double part = 0.0;
            for (int i = lane; i < R; i += 64) part += rs[i];
            for (int j = lane; j < Cn; j += 64) part += ct[j];
            const double total1d = wave_sum_f64(part) + fbest;
            if (lane == 0) { out[pr] = total1d; status[pr] = 0; }
            return;
        }
    }

    // per-lane state, all in registers: column j = lane + 64*c holds v, minv, way, used, prow;
    // row i = lane + 64*c holds its dual u and the "row is in the alternating tree" flag.
    // The Dijkstra step therefore touches LDS only for the cost row (one read per column slot).
    double v[CW], minv[CW], u[CW];
    int prow[CW], way[CW];
    bool used[CW], rowin[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) { v[c] = 0.0; u[c] = 0.0; prow[c] = -1; way[c] = -1; }

    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    bool failed = false;
    int nsteps = 0;
    (void)nsteps;
    // ---- row-reduction start: u_i = min_j g_ij, v = 0 is dual feasible; every row whose arg-min
    // column is not claimed by a lower row is assigned at once (tight pair), the rest augment ----
    bool rowdone[CW];
    {
        int myarg[CW];
        for (int j = lane; j < Cn; j += 64) owner[j] = 0x7fffffff;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int r = lane + 64 * c;
            myarg[c] = -1; rowdone[c] = false;
            if (c < cw_used && r < R) {
                // scan starts at column r (rotated): rows whose gains tie (typically all zero: the
                // point prefers the diagonal) then claim DIFFERENT columns instead of all column 0
                double mn = INF; int arg = 0;
                int j = r < Cn ? r : 0;
#pragma unroll 4
                for (int t = 0; t < Cn; ++t) {
                    const double g = gain(r, j);
                    if (g < mn) { mn = g; arg = j; }
                    j = (j + 1 == Cn) ? 0 : j + 1;
                }
                u[c] = mn; myarg[c] = arg;
                atomicMin(&owner[arg], r);
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            if (myarg[c] >= 0) rowdone[c] = owner[myarg[c]] == lane + 64 * c;
            const int j = lane + 64 * c;
            if (c < cw_used && j < Cn) { const int o = owner[j]; prow[c] = (o != 0x7fffffff) ? o : -1; }
        }
    }
    for (int i = 0; i < R && !failed; ++i) {
        {
            int dn = 0;
#pragma unroll
            for (int c = 0; c < CW; ++c)
                if (c == (i >> 6)) dn = (int)rl32((u32)(rowdone[c] ? 1 : 0), i & 63);
            if (dn) continue;
        }
#pragma unroll
        for (int c = 0; c < CW; ++c) { minv[c] = INF; used[c] = false; rowin[c] = (lane + 64 * c) == i; }
        int i0 = i, j0 = -1;            // j0 = -1 is the virtual start column holding row i
        int steps = 0;
        while (true) {
            double ui0 = 0.0;
#pragma unroll
            for (int c = 0; c < CW; ++c)
                if (c == (i0 >> 6)) ui0 = uni_f64(u[c], i0 & 63);
            // branch-free column update: all cost reads are issued first, invalid / used columns
            // carry +inf candidates
            double best = INF, gg[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                if (c >= cw_used) break;
                const int j = lane + 64 * c;
                const int jc = j < Cn ? j : Cn - 1;
                gg[c] = gain(i0, jc);
            }
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                if (c >= cw_used) break;
                const bool open = (lane + 64 * c) < Cn && !used[c];
                const double cur = gg[c] - ui0 - v[c];
                const bool upd = open && cur < minv[c];
                minv[c] = upd ? cur : minv[c];
                way[c] = upd ? j0 : way[c];
                const double cnd = open ? minv[c] : INF;
                best = cnd < best ? cnd : best;
            }
            const double delta = wave_min_f64_dpp(best);
            ++nsteps;
            // a column attaining delta; ties go to an unassigned column (ends the path at once),
            // the rule scipy's linear_sum_assignment uses too
            int j1 = -1, j1free = -1;
#pragma unroll
            for (int c = CW - 1; c >= 0; --c) {
                if (c >= cw_used) continue;
                const int j = lane + 64 * c;
                const bool at = j < Cn && !used[c] && minv[c] == delta;
                const u64 bal = __ballot(at);
                const u64 balf = __ballot(at && prow[c] < 0);
                if (bal) j1 = 64 * c + __builtin_ctzll(bal);
                if (balf) j1free = 64 * c + __builtin_ctzll(balf);
            }
            if (j1free >= 0) j1 = j1free;
            if (j1 < 0 || !(delta < INF) || ++steps > Cn + 2) { failed = true; break; }
            // dual update: rows in the tree, used / unused columns
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                if (c >= cw_used) break;
                if (rowin[c]) u[c] += delta;
                if (used[c]) v[c] -= delta;
                else minv[c] -= delta;
            }
            // mark j1 used, continue from its row
            const int c1 = j1 >> 6, l1 = j1 & 63;
            int p1 = -1;
#pragma unroll
            for (int c = 0; c < CW; ++c)
                if (c == c1) {
                    p1 = (int)rl32((u32)prow[c], l1);
                    if (lane == l1) used[c] = true;
                }
            j0 = j1;
            if (p1 < 0) break;          // free column reached: augment
            i0 = p1;
#pragma unroll
            for (int c = 0; c < CW; ++c)
                if (c == (p1 >> 6) && lane == (p1 & 63)) rowin[c] = true;
        }
        if (failed) break;
        // augment along way[] back to the virtual column
        int guard = 0;
        while (j0 >= 0 && guard++ < Cn + 2) {
            const int c0 = j0 >> 6, l0 = j0 & 63;
            int jprev = -1;
#pragma unroll
            for (int c = 0; c < CW; ++c)
                if (c == c0) jprev = (int)rl32((u32)way[c], l0);
            int newrow = i;
            if (jprev >= 0) {
                const int cp = jprev >> 6, lp = jprev & 63;
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c == cp) newrow = (int)rl32((u32)prow[c], lp);
            }
#pragma unroll
            for (int c = 0; c < CW; ++c)
                if (c == c0 && lane == l0) prow[c] = newrow;
            j0 = jprev;
        }
    }
    if (failed) {
        if (lane == 0) { out[pr] = __longlong_as_double(0x7ff8000000000000ll); status[pr] = TDA_WIN_NOT_CONVERGED; }
        return;
    }
    WPROF(1, WCLK() - wt0); wt0 = WCLK();
    WPROF(3, nsteps);
    // total = sum over real matches of C_ij + unmatched rows' s + unmatched cols' t
    double part = 0.0;
    __syncthreads();
    // reuse ru[] as "row is really matched" flag
    for (int i = lane; i < R; i += 64) ru[i] = 0.0;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const int j = lane + 64 * c;
        if (j < Cn) {
            bool real = false;
            if (prow[c] >= 0) {
                const double cf = cfull(prow[c], j);
                if (cf - rs[prow[c]] - ct[j] < 0.0) { real = true; part += cf; ru[prow[c]] = 1.0; }
            }
            if (!real) part += ct[j];
        }
    }
    __syncthreads();
    for (int i = lane; i < R; i += 64)
        if (ru[i] == 0.0) part += rs[i];
    const double total = wave_sum_f64(part);
    if (lane == 0) { out[pr] = total; status[pr] = 0; }
    WPROF(2, WCLK() - wt0);
}

// ---------------------------------------------------------------------------------
tda_status launch_wasserstein(tda_ctx* ctx, const double* dgm_a, const int* cnt_a, int cap_a, const double* dgm_b,
                              const int* cnt_b, int cap_b, const int* idx_a, const int* idx_b, int n_pairs,
                              double* out, int* status, hipStream_t st)
{
    if (n_pairs == 0) return TDA_OK;
    if (cap_a < 1 || cap_b < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "diagram capacity must be >= 1");
    const int lo = cap_a < cap_b ? cap_a : cap_b, hi = cap_a < cap_b ? cap_b : cap_a;
    const int max_rows = lo, max_cols = hi;
    if (max_cols > 512) TDA_FAIL(ctx, TDA_ERR_UNSUPPORTED, "diagrams with more than 512 rows are not supported");
    // No cost matrix in LDS: gains are evaluated where they are needed.  A per-pair matrix (16 KB for a 45 x 45 H1
    // pair) left room for 5-7 one-wave workgroups per CU; without it the H0 pairs of a pass take 1.7 instead of 2.9 ms
    // and the H1 pairs 0.73 instead of 1.17 ms (latency-bound solver: residency matters more than the re-evaluation).
    const size_t lds = (size_t)(4 * max_rows + 4 * max_cols + ((max_cols + 1) >> 1)) * 8;
    // diagram buffers with room for more than 128 rows (H1: 256) are mostly far from full (35 x 41 rows on the bench's
    // windows): a first launch with LDS for 64 x 64 points, the launch sized by the capacities for the pairs it defers
    static const bool small_first = getenv("TDA_WS_ONE_LAUNCH") == nullptr;
    const int mode = (small_first && max_cols > 128) ? 2 : 0;
    if (mode) {
        const int sr = max_rows < 64 ? max_rows : 64;
        const size_t lds_s = (size_t)(4 * sr + 4 * 64 + 32) * 8;
        hipLaunchKernelGGL(wasserstein_kernel<1>, dim3(n_pairs), dim3(64), lds_s, st, dgm_a, cnt_a, cap_a, dgm_b, cnt_b, cap_b,
                           idx_a, idx_b, n_pairs, sr, 64, out, status, 1);
    }
#define WS_LAUNCH(CWV)                                                                                         \
    do {                                                                                                       \
        auto kern = wasserstein_kernel<CWV>;                                                                   \
        if (lds > 48 * 1024)                                                                                   \
            TDA_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
        hipLaunchKernelGGL(kern, dim3(n_pairs), dim3(64), lds, st, dgm_a, cnt_a, cap_a, dgm_b, cnt_b, cap_b,   \
                           idx_a, idx_b, n_pairs, max_rows, max_cols, out, status, mode);                      \
    } while (0)
    if (max_cols <= 128) WS_LAUNCH(2);
    else if (max_cols <= 256) WS_LAUNCH(4);
    else WS_LAUNCH(8);
#undef WS_LAUNCH
    TDA_HIP(ctx, hipGetLastError());
    return TDA_OK;
}

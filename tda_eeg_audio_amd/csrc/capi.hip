// capi.hip -- the C ABI of libtdaeeg.so (include/tdaeeg.h): context, argument checks,
// host-pointer twins (stage -> launch -> copy back) and timing helpers.
#include "common.h"
#include <string.h>
#include <vector>

static thread_local std::string g_create_err;

#pragma GCC visibility push(default)
extern "C" {

int tda_version(void) { return 100; }

tda_status tda_ctx_create(int device_id, tda_ctx** out)
{
    if (!out) return TDA_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("no HIP device: ") + hipGetErrorString(e);
        return TDA_ERR_HIP;
    }
    if (device_id < 0 || device_id >= ndev) { g_create_err = "device_id out of range"; return TDA_ERR_INVALID; }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return TDA_ERR_HIP; }
    tda_ctx* c = new (std::nothrow) tda_ctx();
    if (!c) return TDA_ERR_NOMEM;
    c->device = device_id;
    // scratch of the last rung of the Rips ladders, allocated here so that every entry point stays enqueue-only
    e = hipMalloc((void**)&c->total_scratch, rips_total_scratch_bytes());
    if (e != hipSuccess) { g_create_err = std::string("hipMalloc (Rips scratch): ") + hipGetErrorString(e); delete c; return TDA_ERR_HIP; }
    if (retry_lists_reserve(c, 1 << 19) != TDA_OK) { g_create_err = "hipMalloc (retry lists): " + c->err; tda_ctx_destroy(c); return TDA_ERR_HIP; }
    *out = c;
    return TDA_OK;
}

void tda_ctx_destroy(tda_ctx* ctx)
{
    if (!ctx) return;
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->total_scratch) (void)hipFree(ctx->total_scratch);
    for (int i = 0; i < TDA_RETRY_SLOTS; ++i) if (ctx->retry_buf[i]) (void)hipFree(ctx->retry_buf[i]);
    for (void* q : ctx->retired) (void)hipFree(q);
    delete ctx;
}

size_t tda_last_error(const tda_ctx* ctx, char* buf, size_t cap)
{
    const std::string& s = ctx ? ctx->err : g_create_err;
    if (buf && cap) {
        size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}

tda_status tda_set_class_words(tda_ctx* ctx, int words_dm, int words_cloud)
{
    if (!ctx) return TDA_ERR_INVALID;
    if ((words_dm != 0 && words_dm != 1 && words_dm != 2 && words_dm != 4) || (words_cloud != 1 && words_cloud != 2))
        TDA_FAIL(ctx, TDA_ERR_INVALID, "class words: dm in {0,1,2,4}, cloud in {1,2}");
    ctx->words_dm = words_dm;
    ctx->words_cloud = words_cloud;
    return TDA_OK;
}

tda_status tda_set_h1_order(tda_ctx* ctx, int policy)
{
    if (!ctx) return TDA_ERR_INVALID;
    if (policy != TDA_ORDER_IN_CALL && policy != TDA_ORDER_DEFERRED) TDA_FAIL(ctx, TDA_ERR_INVALID, "unknown order policy");
    ctx->h1_order = policy;
    return TDA_OK;
}

tda_status tda_diagram_finish_dev(tda_ctx* ctx, const tda_diagram_set* sets, int n_sets, int n_dgm, void* stream)
{
    if (!ctx) return TDA_ERR_INVALID;
    if (n_dgm < 0) TDA_FAIL(ctx, TDA_ERR_INVALID, "negative count");
    if (n_sets && !sets) TDA_FAIL(ctx, TDA_ERR_INVALID, "null pointer");
    return launch_diagram_finish(ctx, sets, n_sets, n_dgm, (hipStream_t)stream);
}

tda_status tda_set_retry_counter(tda_ctx* ctx, void* dev_counters)
{
    if (!ctx) return TDA_ERR_INVALID;
    ctx->retry_ctr = (unsigned long long*)dev_counters;
    return TDA_OK;
}

tda_status tda_set_retry_policy(tda_ctx* ctx, int policy)
{
    if (!ctx) return TDA_ERR_INVALID;
    if (policy < TDA_RETRY_AUTO || policy > TDA_RETRY_LAST_RUNG) TDA_FAIL(ctx, TDA_ERR_INVALID, "unknown retry policy");
    ctx->retry_policy = policy;
    return TDA_OK;
}

#define CHECK_CTX(ctx) do { if (!(ctx)) return TDA_ERR_INVALID; } while (0)
#define CHECK_PTR(ctx, p) do { if (!(p)) TDA_FAIL(ctx, TDA_ERR_INVALID, "null pointer: " #p); } while (0)
#define CHECK_NONNEG(ctx, v) do { if ((v) < 0) TDA_FAIL(ctx, TDA_ERR_INVALID, "negative size: " #v); } while (0)

// ---------------------------------------------------------------- device-pointer API
tda_status tda_corr_dist_batch_dev(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t, double* dist,
                                   double* corr, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win) { CHECK_PTR(ctx, win); CHECK_PTR(ctx, dist); }
    return launch_corr_dist(ctx, win, n_win, n_ch, n_t, dist, corr, (hipStream_t)stream);
}

tda_status tda_corr_dist_sliding_dev(tda_ctx* ctx, const double* sig, int n_ch, int n_samples, int win_len, int step,
                                     double* dist, double* corr, int* n_win, void* stream)
{
    CHECK_CTX(ctx); CHECK_PTR(ctx, sig); CHECK_PTR(ctx, dist);
    CHECK_NONNEG(ctx, n_samples);
    return launch_corr_dist_sliding(ctx, sig, n_ch, n_samples, win_len, step, dist, corr, n_win, (hipStream_t)stream);
}

tda_status tda_corr_to_dist_batch_dev(tda_ctx* ctx, const double* corr, int n_win, int n, int method, double* dist,
                                      void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win) { CHECK_PTR(ctx, corr); CHECK_PTR(ctx, dist); }
    if (n < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "n must be >= 1");
    return launch_corr_to_dist(ctx, corr, n_win, n, method, dist, (hipStream_t)stream);
}

tda_status tda_rips_dm_batch_dev(tda_ctx* ctx, const double* dm, int n_win, int n, double thresh, int symmetrise,
                                 double* h0, int h0_cap, int* h0_cnt, double* h1, int h1_cap, int* h1_cnt,
                                 int* status, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win) { CHECK_PTR(ctx, dm); CHECK_PTR(ctx, h0); CHECK_PTR(ctx, h0_cnt); CHECK_PTR(ctx, h1);
                 CHECK_PTR(ctx, h1_cnt); CHECK_PTR(ctx, status); }
    if (h1_cap < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "h1_cap must be >= 1");
    return launch_rips_dm(ctx, dm, n_win, n, thresh, symmetrise, h0, h0_cap, h0_cnt, h1, h1_cap, h1_cnt, status,
                          (hipStream_t)stream);
}

tda_status tda_eeg_window_batch_dev(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t, double thresh,
                                    double* dist, double* corr, double* h0, int h0_cap, int* h0_cnt, double* h1,
                                    int h1_cap, int* h1_cnt, int* status, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win) { CHECK_PTR(ctx, win); CHECK_PTR(ctx, h0); CHECK_PTR(ctx, h0_cnt); CHECK_PTR(ctx, h1);
                 CHECK_PTR(ctx, h1_cnt); CHECK_PTR(ctx, status); }
    if (corr && !dist) TDA_FAIL(ctx, TDA_ERR_INVALID, "corr needs dist");
    if (h1_cap < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "h1_cap must be >= 1");
    return launch_eeg_windows(ctx, win, n_win, n_ch, n_t, thresh, dist, corr, h0, h0_cap, h0_cnt, h1, h1_cap, h1_cnt,
                              status, (hipStream_t)stream);
}

tda_status tda_eeg_window_sliding_dev(tda_ctx* ctx, const double* sig, int n_rec, int n_ch, int n_samples, int win_len,
                                      int step, const int* sel, int n_sel, double thresh, double* dist, double* corr,
                                      double* h0, int h0_cap, int* h0_cnt, double* h1, int h1_cap, int* h1_cnt,
                                      int* status, int* n_win_per_rec, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_rec); CHECK_NONNEG(ctx, n_sel);
    if (n_rec) { CHECK_PTR(ctx, sig); CHECK_PTR(ctx, h0); CHECK_PTR(ctx, h0_cnt); CHECK_PTR(ctx, h1);
                 CHECK_PTR(ctx, h1_cnt); CHECK_PTR(ctx, status); }
    if (corr && !dist) TDA_FAIL(ctx, TDA_ERR_INVALID, "corr needs dist");
    if (h1_cap < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "h1_cap must be >= 1");
    return launch_eeg_sliding(ctx, sig, n_rec, n_ch, n_samples, win_len, step, sel, n_sel, thresh, dist, corr, h0, h0_cap,
                              h0_cnt, h1, h1_cap, h1_cnt, status, n_win_per_rec, (hipStream_t)stream);
}

tda_status tda_takens_rips_batch_dev(tda_ctx* ctx, const double* win, const int* tau, int n_win, int n_t, int dim,
                                     int subsample, double thresh, double* h0, int h0_cap, int* h0_cnt, double* h1,
                                     int h1_cap, int* h1_cnt, int* n_points, int* status, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win) { CHECK_PTR(ctx, win); CHECK_PTR(ctx, tau); CHECK_PTR(ctx, h0); CHECK_PTR(ctx, h0_cnt);
                 CHECK_PTR(ctx, h1); CHECK_PTR(ctx, h1_cnt); CHECK_PTR(ctx, status); }
    if (h1_cap < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "h1_cap must be >= 1");
    if (n_t < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "n_t must be >= 1");
    return launch_rips_cloud(ctx, win, tau, n_win, n_t, dim, subsample, 0, 1, thresh, h0, h0_cap, h0_cnt, h1, h1_cap,
                             h1_cnt, n_points, status, (hipStream_t)stream);
}

tda_status tda_cloud_rips_batch_dev(tda_ctx* ctx, const double* pc, const int* n_pts, int n_win, int p_cap, int dim,
                                    int normalise, double thresh, double* h0, int h0_cap, int* h0_cnt, double* h1,
                                    int h1_cap, int* h1_cnt, int* status, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win) { CHECK_PTR(ctx, pc); CHECK_PTR(ctx, n_pts); CHECK_PTR(ctx, h0); CHECK_PTR(ctx, h0_cnt);
                 CHECK_PTR(ctx, h1); CHECK_PTR(ctx, h1_cnt); CHECK_PTR(ctx, status); }
    if (h1_cap < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "h1_cap must be >= 1");
    if (p_cap < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "p_cap must be >= 1");
    return launch_rips_cloud(ctx, pc, n_pts, n_win, p_cap, dim, 1, 1, normalise, thresh, h0, h0_cap, h0_cnt, h1,
                             h1_cap, h1_cnt, nullptr, status, (hipStream_t)stream);
}

tda_status tda_sosfiltfilt_dev(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* sos,
                               const double* zi, int n_sections, int edge, double* y, double* work, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_sig);
    if (n_sig) { CHECK_PTR(ctx, x); CHECK_PTR(ctx, y); CHECK_PTR(ctx, work); }
    CHECK_PTR(ctx, sos); CHECK_PTR(ctx, zi);
    return launch_sosfiltfilt(ctx, x, n_sig, n_samples, sos, zi, n_sections, edge, y, work, (hipStream_t)stream);
}

tda_status tda_filtfilt_dev(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* b, const double* a,
                            const double* zi, int ntaps, int edge, double* y, double* work, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_sig);
    if (n_sig) { CHECK_PTR(ctx, x); CHECK_PTR(ctx, y); CHECK_PTR(ctx, work); }
    CHECK_PTR(ctx, b); CHECK_PTR(ctx, a); CHECK_PTR(ctx, zi);
    return launch_filtfilt(ctx, x, n_sig, n_samples, b, a, zi, ntaps, edge, y, work, (hipStream_t)stream);
}

tda_status tda_sosfiltfilt_bank_dev(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* sos,
                                    const double* zi, int n_filters, int n_sections, int edge, double* y, double* work,
                                    void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_sig);
    if (n_sig) { CHECK_PTR(ctx, x); CHECK_PTR(ctx, y); CHECK_PTR(ctx, work); }
    CHECK_PTR(ctx, sos); CHECK_PTR(ctx, zi);
    return launch_sosfiltfilt(ctx, x, n_sig, n_samples, sos, zi, n_sections, edge, y, work, (hipStream_t)stream, n_filters);
}

tda_status tda_filtfilt_bank_dev(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* b, const double* a,
                                 const double* zi, int n_filters, int ntaps, int edge, double* y, double* work, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_sig);
    if (n_sig) { CHECK_PTR(ctx, x); CHECK_PTR(ctx, y); CHECK_PTR(ctx, work); }
    CHECK_PTR(ctx, b); CHECK_PTR(ctx, a); CHECK_PTR(ctx, zi);
    return launch_filtfilt(ctx, x, n_sig, n_samples, b, a, zi, ntaps, edge, y, work, (hipStream_t)stream, n_filters);
}

tda_status tda_upfirdn_dev(tda_ctx* ctx, const double* x, long long n_in, const double* h, int len_h, int up, int down,
                           long long n_pre_remove, long long n_out, double* y, void* stream)
{
    CHECK_CTX(ctx); CHECK_PTR(ctx, x); CHECK_PTR(ctx, h); CHECK_PTR(ctx, y);
    return launch_upfirdn(ctx, x, n_in, h, len_h, up, down, n_pre_remove, n_out, y, (hipStream_t)stream);
}

tda_status tda_hilbert_envelope_dev(tda_ctx* ctx, const double* x, int n, const double* g, double* env, void* stream)
{
    CHECK_CTX(ctx); CHECK_PTR(ctx, x); CHECK_PTR(ctx, g); CHECK_PTR(ctx, env);
    return launch_hilbert_env(ctx, x, n, g, env, (hipStream_t)stream);
}

tda_status tda_tau_batch_dev(tda_ctx* ctx, const double* win, int n_win, int n_t, int max_lag, int* tau, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win) { CHECK_PTR(ctx, win); CHECK_PTR(ctx, tau); }
    return launch_tau(ctx, win, n_win, n_t, max_lag, tau, (hipStream_t)stream);
}

tda_status tda_tau_segments_dev(tda_ctx* ctx, const double* win, const int* seg_off, int n_seg, int n_t, int max_lag,
                                int* tau_seg, int* tau_win, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_seg);
    if (n_seg) { CHECK_PTR(ctx, win); CHECK_PTR(ctx, seg_off); CHECK_PTR(ctx, tau_seg); }
    return launch_tau_segments(ctx, win, seg_off, n_seg, n_t, max_lag, tau_seg, tau_win, (hipStream_t)stream);
}

tda_status tda_recording_rows_dev(tda_ctx* ctx, const double* w_h0, const double* w_h1, const int* tau_seg,
                                  const double* feat_h0, const double* feat_h1, const int* seg_off, int n_seg,
                                  double* out, const int* status_a, const int* status_b, int* seg_flags, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_seg);
    if (n_seg) { CHECK_PTR(ctx, w_h0); CHECK_PTR(ctx, w_h1); CHECK_PTR(ctx, tau_seg); CHECK_PTR(ctx, feat_h0);
                 CHECK_PTR(ctx, feat_h1); CHECK_PTR(ctx, seg_off); CHECK_PTR(ctx, out); }
    return launch_recording_rows(ctx, w_h0, w_h1, tau_seg, feat_h0, feat_h1, seg_off, n_seg, out, status_a, status_b,
                                 seg_flags, (hipStream_t)stream);
}

tda_status tda_features_batch_dev(tda_ctx* ctx, const double* dgm, const int* cnt, int n_dgm, int cap, double* feat,
                                  void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_dgm);
    if (n_dgm) { CHECK_PTR(ctx, dgm); CHECK_PTR(ctx, cnt); CHECK_PTR(ctx, feat); }
    return launch_features(ctx, dgm, cnt, n_dgm, cap, feat, (hipStream_t)stream);
}

tda_status tda_aggregate_batch_dev(tda_ctx* ctx, const double* feat_h0, const double* feat_h1, const int* seg_off,
                                   int n_seg, double* out, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_seg);
    if (n_seg) { CHECK_PTR(ctx, feat_h0); CHECK_PTR(ctx, feat_h1); CHECK_PTR(ctx, seg_off); CHECK_PTR(ctx, out); }
    return launch_aggregate(ctx, feat_h0, feat_h1, seg_off, n_seg, out, (hipStream_t)stream);
}

tda_status tda_segment_nanmean_dev(tda_ctx* ctx, const double* x, const int* seg_off, int n_seg, double* out,
                                   void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_seg);
    if (n_seg) { CHECK_PTR(ctx, x); CHECK_PTR(ctx, seg_off); CHECK_PTR(ctx, out); }
    return launch_nanmean(ctx, x, seg_off, n_seg, out, (hipStream_t)stream);
}

tda_status tda_spearman_batch_dev(tda_ctx* ctx, const double* x, const double* y, int ld, const int* cols, int n_cols,
                                  const int* seg_off, int n_seg, double* r, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_seg);
    if (n_seg) { CHECK_PTR(ctx, x); CHECK_PTR(ctx, y); CHECK_PTR(ctx, cols); CHECK_PTR(ctx, seg_off); CHECK_PTR(ctx, r); }
    return launch_spearman(ctx, x, y, ld, cols, n_cols, seg_off, n_seg, r, (hipStream_t)stream);
}

tda_status tda_wasserstein_batch_dev(tda_ctx* ctx, const double* dgm_a, const int* cnt_a, int cap_a,
                                     const double* dgm_b, const int* cnt_b, int cap_b, const int* idx_a,
                                     const int* idx_b, int n_pairs, double* out, int* status, void* stream)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_pairs);
    if (n_pairs) { CHECK_PTR(ctx, dgm_a); CHECK_PTR(ctx, cnt_a); CHECK_PTR(ctx, dgm_b); CHECK_PTR(ctx, cnt_b);
                   CHECK_PTR(ctx, out); CHECK_PTR(ctx, status); }
    return launch_wasserstein(ctx, dgm_a, cnt_a, cap_a, dgm_b, cnt_b, cap_b, idx_a, idx_b, n_pairs, out, status,
                              (hipStream_t)stream);
}

// ---------------------------------------------------------------- host-pointer twins
// A bump allocator over the context workspace; everything is staged, launched on the
// default stream, copied back and synchronised.
struct Stage {
    tda_ctx* ctx;
    size_t need = 0, used = 0;
    struct Item { void** dev; const void* src; void* dst; size_t bytes; };
    std::vector<Item> items;
    explicit Stage(tda_ctx* c) : ctx(c) {}
    void add(void** dev, const void* src, void* dst, size_t bytes)
    {
        items.push_back({dev, src, dst, bytes});
        need += (bytes + 255) & ~(size_t)255;
    }
    tda_status upload()
    {
        if (need > ctx->ws_bytes) {
            if (ctx->ws) TDA_HIP(ctx, hipFree(ctx->ws));
            ctx->ws = nullptr; ctx->ws_bytes = 0;
            TDA_HIP(ctx, hipMalloc(&ctx->ws, need));
            ctx->ws_bytes = need;
        }
        char* p = (char*)ctx->ws;
        for (auto& it : items) {
            *it.dev = it.bytes ? p : nullptr;
            if (it.src && it.bytes) TDA_HIP(ctx, hipMemcpyAsync(p, it.src, it.bytes, hipMemcpyHostToDevice, 0));
            p += (it.bytes + 255) & ~(size_t)255;
        }
        return TDA_OK;
    }
    tda_status download()
    {
        for (auto& it : items)
            if (it.dst && it.bytes) TDA_HIP(ctx, hipMemcpyAsync(it.dst, *it.dev, it.bytes, hipMemcpyDeviceToHost, 0));
        TDA_HIP(ctx, hipStreamSynchronize(0));
        return TDA_OK;
    }
};

#define RET_IF(x) do { tda_status s__ = (x); if (s__ != TDA_OK) return s__; } while (0)

tda_status tda_corr_dist_batch(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t, double* dist, double* corr)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win == 0) return TDA_OK;
    CHECK_PTR(ctx, win); CHECK_PTR(ctx, dist);
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_win, *d_dist, *d_corr = nullptr;
    const size_t nw = (size_t)n_win;
    s.add((void**)&d_win, win, nullptr, nw * n_ch * n_t * 8);
    s.add((void**)&d_dist, nullptr, dist, nw * n_ch * n_ch * 8);
    if (corr) s.add((void**)&d_corr, nullptr, corr, nw * n_ch * n_ch * 8);
    RET_IF(s.upload());
    RET_IF(tda_corr_dist_batch_dev(ctx, d_win, n_win, n_ch, n_t, d_dist, d_corr, nullptr));
    return s.download();
}

tda_status tda_corr_dist_sliding(tda_ctx* ctx, const double* sig, int n_ch, int n_samples, int win_len, int step,
                                 double* dist, double* corr, int* n_win)
{
    CHECK_CTX(ctx); CHECK_PTR(ctx, sig); CHECK_PTR(ctx, dist); CHECK_NONNEG(ctx, n_samples);
    if (win_len < 2 || step < 1 || n_ch < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "bad window geometry");
    const int nw = n_samples >= win_len ? (n_samples - win_len) / step + 1 : 0;
    if (n_win) *n_win = nw;
    if (nw == 0) return TDA_OK;
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_sig, *d_dist, *d_corr = nullptr;
    s.add((void**)&d_sig, sig, nullptr, (size_t)n_ch * n_samples * 8);
    s.add((void**)&d_dist, nullptr, dist, (size_t)nw * n_ch * n_ch * 8);
    if (corr) s.add((void**)&d_corr, nullptr, corr, (size_t)nw * n_ch * n_ch * 8);
    RET_IF(s.upload());
    RET_IF(tda_corr_dist_sliding_dev(ctx, d_sig, n_ch, n_samples, win_len, step, d_dist, d_corr, nullptr, nullptr));
    return s.download();
}

tda_status tda_corr_to_dist_batch(tda_ctx* ctx, const double* corr, int n_win, int n, int method, double* dist)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win == 0) return TDA_OK;
    CHECK_PTR(ctx, corr); CHECK_PTR(ctx, dist);
    if (n < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "n must be >= 1");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_c, *d_d;
    s.add((void**)&d_c, corr, nullptr, (size_t)n_win * n * n * 8);
    s.add((void**)&d_d, nullptr, dist, (size_t)n_win * n * n * 8);
    RET_IF(s.upload());
    RET_IF(tda_corr_to_dist_batch_dev(ctx, d_c, n_win, n, method, d_d, nullptr));
    return s.download();
}

tda_status tda_rips_dm_batch(tda_ctx* ctx, const double* dm, int n_win, int n, double thresh, int symmetrise,
                             double* h0, int h0_cap, int* h0_cnt, double* h1, int h1_cap, int* h1_cnt, int* status)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win == 0) return TDA_OK;
    CHECK_PTR(ctx, dm); CHECK_PTR(ctx, h0); CHECK_PTR(ctx, h0_cnt); CHECK_PTR(ctx, h1); CHECK_PTR(ctx, h1_cnt);
    CHECK_PTR(ctx, status);
    if (h0_cap < 1 || h1_cap < 1 || n < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "n, h0_cap, h1_cap must be >= 1");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_dm, *d_h0, *d_h1; int *d_c0, *d_c1, *d_st;
    const size_t nw = (size_t)n_win;
    s.add((void**)&d_dm, dm, nullptr, nw * n * n * 8);
    s.add((void**)&d_h0, nullptr, h0, nw * h0_cap * 16);
    s.add((void**)&d_h1, nullptr, h1, nw * h1_cap * 16);
    s.add((void**)&d_c0, nullptr, h0_cnt, nw * 4);
    s.add((void**)&d_c1, nullptr, h1_cnt, nw * 4);
    s.add((void**)&d_st, nullptr, status, nw * 4);
    RET_IF(s.upload());
    RET_IF(tda_rips_dm_batch_dev(ctx, d_dm, n_win, n, thresh, symmetrise, d_h0, h0_cap, d_c0, d_h1, h1_cap, d_c1, d_st,
                                 nullptr));
    return s.download();
}

tda_status tda_takens_rips_batch(tda_ctx* ctx, const double* win, const int* tau, int n_win, int n_t, int dim,
                                 int subsample, double thresh, double* h0, int h0_cap, int* h0_cnt, double* h1,
                                 int h1_cap, int* h1_cnt, int* n_points, int* status)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win == 0) return TDA_OK;
    CHECK_PTR(ctx, win); CHECK_PTR(ctx, tau); CHECK_PTR(ctx, h0); CHECK_PTR(ctx, h0_cnt); CHECK_PTR(ctx, h1);
    CHECK_PTR(ctx, h1_cnt); CHECK_PTR(ctx, status);
    if (h0_cap < 1 || h1_cap < 1 || n_t < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "n_t, h0_cap, h1_cap must be >= 1");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_win, *d_h0, *d_h1; int *d_tau, *d_c0, *d_c1, *d_np, *d_st;
    const size_t nw = (size_t)n_win;
    s.add((void**)&d_win, win, nullptr, nw * n_t * 8);
    s.add((void**)&d_tau, tau, nullptr, nw * 4);
    s.add((void**)&d_h0, nullptr, h0, nw * h0_cap * 16);
    s.add((void**)&d_h1, nullptr, h1, nw * h1_cap * 16);
    s.add((void**)&d_c0, nullptr, h0_cnt, nw * 4);
    s.add((void**)&d_c1, nullptr, h1_cnt, nw * 4);
    s.add((void**)&d_np, nullptr, n_points, nw * 4);
    s.add((void**)&d_st, nullptr, status, nw * 4);
    RET_IF(s.upload());
    RET_IF(tda_takens_rips_batch_dev(ctx, d_win, d_tau, n_win, n_t, dim, subsample, thresh, d_h0, h0_cap, d_c0, d_h1,
                                     h1_cap, d_c1, d_np, d_st, nullptr));
    return s.download();
}

tda_status tda_cloud_rips_batch(tda_ctx* ctx, const double* pc, const int* n_pts, int n_win, int p_cap, int dim,
                                int normalise, double thresh, double* h0, int h0_cap, int* h0_cnt, double* h1,
                                int h1_cap, int* h1_cnt, int* status)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win == 0) return TDA_OK;
    CHECK_PTR(ctx, pc); CHECK_PTR(ctx, n_pts); CHECK_PTR(ctx, h0); CHECK_PTR(ctx, h0_cnt); CHECK_PTR(ctx, h1);
    CHECK_PTR(ctx, h1_cnt); CHECK_PTR(ctx, status);
    if (h0_cap < 1 || h1_cap < 1 || p_cap < 1 || dim < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "sizes must be >= 1");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_pc, *d_h0, *d_h1; int *d_n, *d_c0, *d_c1, *d_st;
    const size_t nw = (size_t)n_win;
    s.add((void**)&d_pc, pc, nullptr, nw * p_cap * dim * 8);
    s.add((void**)&d_n, n_pts, nullptr, nw * 4);
    s.add((void**)&d_h0, nullptr, h0, nw * h0_cap * 16);
    s.add((void**)&d_h1, nullptr, h1, nw * h1_cap * 16);
    s.add((void**)&d_c0, nullptr, h0_cnt, nw * 4);
    s.add((void**)&d_c1, nullptr, h1_cnt, nw * 4);
    s.add((void**)&d_st, nullptr, status, nw * 4);
    RET_IF(s.upload());
    RET_IF(tda_cloud_rips_batch_dev(ctx, d_pc, d_n, n_win, p_cap, dim, normalise, thresh, d_h0, h0_cap, d_c0, d_h1,
                                    h1_cap, d_c1, d_st, nullptr));
    return s.download();
}

tda_status tda_sosfiltfilt(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* sos, const double* zi,
                           int n_sections, int edge, double* y)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_sig);
    if (n_sig == 0) return TDA_OK;
    CHECK_PTR(ctx, x); CHECK_PTR(ctx, y); CHECK_PTR(ctx, sos); CHECK_PTR(ctx, zi);
    if (n_samples < 1 || edge < 0) TDA_FAIL(ctx, TDA_ERR_INVALID, "bad sizes");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_x, *d_y, *d_w;
    s.add((void**)&d_x, x, nullptr, (size_t)n_sig * n_samples * 8);
    s.add((void**)&d_y, nullptr, y, (size_t)n_sig * n_samples * 8);
    s.add((void**)&d_w, nullptr, nullptr, (size_t)n_sig * (n_samples + 2 * edge) * 8);
    RET_IF(s.upload());
    RET_IF(tda_sosfiltfilt_dev(ctx, d_x, n_sig, n_samples, sos, zi, n_sections, edge, d_y, d_w, nullptr));
    return s.download();
}

tda_status tda_filtfilt(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* b, const double* a,
                        const double* zi, int ntaps, int edge, double* y)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_sig);
    if (n_sig == 0) return TDA_OK;
    CHECK_PTR(ctx, x); CHECK_PTR(ctx, y); CHECK_PTR(ctx, b); CHECK_PTR(ctx, a); CHECK_PTR(ctx, zi);
    if (n_samples < 1 || edge < 0) TDA_FAIL(ctx, TDA_ERR_INVALID, "bad sizes");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_x, *d_y, *d_w;
    s.add((void**)&d_x, x, nullptr, (size_t)n_sig * n_samples * 8);
    s.add((void**)&d_y, nullptr, y, (size_t)n_sig * n_samples * 8);
    s.add((void**)&d_w, nullptr, nullptr, (size_t)n_sig * (n_samples + 2 * edge) * 8);
    RET_IF(s.upload());
    RET_IF(tda_filtfilt_dev(ctx, d_x, n_sig, n_samples, b, a, zi, ntaps, edge, d_y, d_w, nullptr));
    return s.download();
}

tda_status tda_upfirdn(tda_ctx* ctx, const double* x, long long n_in, const double* h, int len_h, int up, int down,
                       long long n_pre_remove, long long n_out, double* y)
{
    CHECK_CTX(ctx); CHECK_PTR(ctx, x); CHECK_PTR(ctx, h); CHECK_PTR(ctx, y);
    if (n_in < 1 || n_out < 1 || len_h < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "bad sizes");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_x, *d_h, *d_y;
    s.add((void**)&d_x, x, nullptr, (size_t)n_in * 8);
    s.add((void**)&d_h, h, nullptr, (size_t)len_h * 8);
    s.add((void**)&d_y, nullptr, y, (size_t)n_out * 8);
    RET_IF(s.upload());
    RET_IF(tda_upfirdn_dev(ctx, d_x, n_in, d_h, len_h, up, down, n_pre_remove, n_out, d_y, nullptr));
    return s.download();
}

tda_status tda_hilbert_envelope(tda_ctx* ctx, const double* x, int n, const double* g, double* env)
{
    CHECK_CTX(ctx); CHECK_PTR(ctx, x); CHECK_PTR(ctx, g); CHECK_PTR(ctx, env);
    if (n < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "bad sizes");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_x, *d_g, *d_e;
    s.add((void**)&d_x, x, nullptr, (size_t)n * 8);
    s.add((void**)&d_g, g, nullptr, (size_t)n * 8);
    s.add((void**)&d_e, nullptr, env, (size_t)n * 8);
    RET_IF(s.upload());
    RET_IF(tda_hilbert_envelope_dev(ctx, d_x, n, d_g, d_e, nullptr));
    return s.download();
}

tda_status tda_tau_batch(tda_ctx* ctx, const double* win, int n_win, int n_t, int max_lag, int* tau)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_win);
    if (n_win == 0) return TDA_OK;
    CHECK_PTR(ctx, win); CHECK_PTR(ctx, tau);
    if (n_t < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "n_t must be >= 1");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double* d_win; int* d_tau;
    s.add((void**)&d_win, win, nullptr, (size_t)n_win * n_t * 8);
    s.add((void**)&d_tau, nullptr, tau, (size_t)n_win * 4);
    RET_IF(s.upload());
    RET_IF(tda_tau_batch_dev(ctx, d_win, n_win, n_t, max_lag, d_tau, nullptr));
    return s.download();
}

tda_status tda_features_batch(tda_ctx* ctx, const double* dgm, const int* cnt, int n_dgm, int cap, double* feat)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_dgm);
    if (n_dgm == 0) return TDA_OK;
    CHECK_PTR(ctx, dgm); CHECK_PTR(ctx, cnt); CHECK_PTR(ctx, feat);
    if (cap < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "cap must be >= 1");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_dgm, *d_feat; int* d_cnt;
    s.add((void**)&d_dgm, dgm, nullptr, (size_t)n_dgm * cap * 16);
    s.add((void**)&d_cnt, cnt, nullptr, (size_t)n_dgm * 4);
    s.add((void**)&d_feat, nullptr, feat, (size_t)n_dgm * TDA_N_FEATURES * 8);
    RET_IF(s.upload());
    RET_IF(tda_features_batch_dev(ctx, d_dgm, d_cnt, n_dgm, cap, d_feat, nullptr));
    return s.download();
}

tda_status tda_aggregate_batch(tda_ctx* ctx, const double* feat_h0, const double* feat_h1, const int* seg_off,
                               int n_seg, int n_total, double* out)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_seg); CHECK_NONNEG(ctx, n_total);
    if (n_seg == 0) return TDA_OK;
    CHECK_PTR(ctx, feat_h0); CHECK_PTR(ctx, feat_h1); CHECK_PTR(ctx, seg_off); CHECK_PTR(ctx, out);
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d0, *d1, *d_out; int* d_off;
    s.add((void**)&d0, feat_h0, nullptr, (size_t)n_total * TDA_N_FEATURES * 8);
    s.add((void**)&d1, feat_h1, nullptr, (size_t)n_total * TDA_N_FEATURES * 8);
    s.add((void**)&d_off, seg_off, nullptr, (size_t)(n_seg + 1) * 4);
    s.add((void**)&d_out, nullptr, out, (size_t)n_seg * 4 * TDA_N_FEATURES * 8);
    RET_IF(s.upload());
    RET_IF(tda_aggregate_batch_dev(ctx, d0, d1, d_off, n_seg, d_out, nullptr));
    return s.download();
}

tda_status tda_segment_nanmean(tda_ctx* ctx, const double* x, const int* seg_off, int n_seg, int n_total, double* out)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_seg); CHECK_NONNEG(ctx, n_total);
    if (n_seg == 0) return TDA_OK;
    CHECK_PTR(ctx, x); CHECK_PTR(ctx, seg_off); CHECK_PTR(ctx, out);
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_x, *d_out; int* d_off;
    s.add((void**)&d_x, x, nullptr, (size_t)n_total * 8);
    s.add((void**)&d_off, seg_off, nullptr, (size_t)(n_seg + 1) * 4);
    s.add((void**)&d_out, nullptr, out, (size_t)n_seg * 8);
    RET_IF(s.upload());
    RET_IF(tda_segment_nanmean_dev(ctx, d_x, d_off, n_seg, d_out, nullptr));
    return s.download();
}

tda_status tda_spearman_batch(tda_ctx* ctx, const double* x, const double* y, int n_total, int ld, const int* cols,
                              int n_cols, const int* seg_off, int n_seg, double* r)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_seg);
    if (n_seg == 0) return TDA_OK;
    CHECK_PTR(ctx, x); CHECK_PTR(ctx, y); CHECK_PTR(ctx, cols); CHECK_PTR(ctx, seg_off); CHECK_PTR(ctx, r);
    if (n_total < 1 || ld < 1 || n_cols < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "sizes must be >= 1");
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_x, *d_y, *d_r; int *d_c, *d_o;
    s.add((void**)&d_x, x, nullptr, (size_t)n_total * ld * 8);
    s.add((void**)&d_y, y, nullptr, (size_t)n_total * ld * 8);
    s.add((void**)&d_c, cols, nullptr, (size_t)n_cols * 4);
    s.add((void**)&d_o, seg_off, nullptr, (size_t)(n_seg + 1) * 4);
    s.add((void**)&d_r, nullptr, r, (size_t)n_seg * n_cols * 8);
    RET_IF(s.upload());
    RET_IF(tda_spearman_batch_dev(ctx, d_x, d_y, ld, d_c, n_cols, d_o, n_seg, d_r, nullptr));
    return s.download();
}

tda_status tda_wasserstein_batch(tda_ctx* ctx, const double* dgm_a, const int* cnt_a, int n_a, int cap_a,
                                 const double* dgm_b, const int* cnt_b, int n_b, int cap_b, const int* idx_a,
                                 const int* idx_b, int n_pairs, double* out, int* status)
{
    CHECK_CTX(ctx); CHECK_NONNEG(ctx, n_pairs);
    if (n_pairs == 0) return TDA_OK;
    CHECK_PTR(ctx, dgm_a); CHECK_PTR(ctx, cnt_a); CHECK_PTR(ctx, dgm_b); CHECK_PTR(ctx, cnt_b); CHECK_PTR(ctx, out);
    CHECK_PTR(ctx, status);
    if (n_a < 1 || n_b < 1 || cap_a < 1 || cap_b < 1) TDA_FAIL(ctx, TDA_ERR_INVALID, "sizes must be >= 1");
    for (int i = 0; i < n_pairs; ++i) {
        const int a = idx_a ? idx_a[i] : i, b = idx_b ? idx_b[i] : i;
        if (a < 0 || a >= n_a || b < 0 || b >= n_b) TDA_FAIL(ctx, TDA_ERR_INVALID, "pair index out of range");
    }
    TDA_HIP(ctx, hipSetDevice(ctx->device));
    Stage s(ctx);
    double *d_a, *d_b, *d_out; int *d_ca, *d_cb, *d_ia = nullptr, *d_ib = nullptr, *d_st;
    s.add((void**)&d_a, dgm_a, nullptr, (size_t)n_a * cap_a * 16);
    s.add((void**)&d_ca, cnt_a, nullptr, (size_t)n_a * 4);
    s.add((void**)&d_b, dgm_b, nullptr, (size_t)n_b * cap_b * 16);
    s.add((void**)&d_cb, cnt_b, nullptr, (size_t)n_b * 4);
    if (idx_a) s.add((void**)&d_ia, idx_a, nullptr, (size_t)n_pairs * 4);
    if (idx_b) s.add((void**)&d_ib, idx_b, nullptr, (size_t)n_pairs * 4);
    s.add((void**)&d_out, nullptr, out, (size_t)n_pairs * 8);
    s.add((void**)&d_st, nullptr, status, (size_t)n_pairs * 4);
    RET_IF(s.upload());
    RET_IF(tda_wasserstein_batch_dev(ctx, d_a, d_ca, cap_a, d_b, d_cb, cap_b, d_ia, d_ib, n_pairs, d_out, d_st, nullptr));
    return s.download();
}

// ---------------------------------------------------------------- timing helpers
tda_status tda_event_create(tda_ctx* ctx, void** ev)
{
    CHECK_CTX(ctx); CHECK_PTR(ctx, ev);
    hipEvent_t e;
    TDA_HIP(ctx, hipEventCreate(&e));
    *ev = (void*)e;
    return TDA_OK;
}
tda_status tda_event_record(tda_ctx* ctx, void* ev, void* stream)
{
    CHECK_CTX(ctx); CHECK_PTR(ctx, ev);
    TDA_HIP(ctx, hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return TDA_OK;
}
tda_status tda_event_elapsed_ms(tda_ctx* ctx, void* ev_start, void* ev_stop, float* ms)
{
    CHECK_CTX(ctx); CHECK_PTR(ctx, ev_start); CHECK_PTR(ctx, ev_stop); CHECK_PTR(ctx, ms);
    TDA_HIP(ctx, hipEventSynchronize((hipEvent_t)ev_stop));
    TDA_HIP(ctx, hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return TDA_OK;
}
tda_status tda_event_destroy(tda_ctx* ctx, void* ev)
{
    CHECK_CTX(ctx);
    if (ev) TDA_HIP(ctx, hipEventDestroy((hipEvent_t)ev));
    return TDA_OK;
}
tda_status tda_set_kernel_probe(tda_ctx* ctx, int which, void* ev_start, void* ev_stop, void* dev_span)
{
    CHECK_CTX(ctx);
    if (which < TDA_PROBE_NONE || which > TDA_PROBE_CORR_DIST) TDA_FAIL(ctx, TDA_ERR_INVALID, "unknown probe");
    if (which != TDA_PROBE_NONE && !(ev_start && ev_stop) && !dev_span)
        TDA_FAIL(ctx, TDA_ERR_INVALID, "probe needs two events or a span buffer");
    if ((ev_start == nullptr) != (ev_stop == nullptr)) TDA_FAIL(ctx, TDA_ERR_INVALID, "probe events come in pairs");
    if (which == TDA_PROBE_NONE) {
        for (int w = 0; w < 4; ++w) ctx->probe[w] = tda_ctx::Probe();
        return TDA_OK;
    }
    ctx->probe[which].start = (hipEvent_t)ev_start;
    ctx->probe[which].stop = (hipEvent_t)ev_stop;
    ctx->probe[which].span = (unsigned long long*)dev_span;
    ctx->probe[which].armed = true;
    return TDA_OK;
}
tda_status tda_stream_sync(tda_ctx* ctx, void* stream)
{
    CHECK_CTX(ctx);
    TDA_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return TDA_OK;
}

}  // extern "C"
#pragma GCC visibility pop

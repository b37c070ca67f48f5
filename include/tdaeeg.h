/*
 * tdaeeg.h -- C ABI of libtdaeeg.so: the MI355X (gfx950) persistent-homology
 * feature engine for the per-window hot path of Ignaciagothe/tda-eeg-audio.
 *
 * The reference has no FFI: its boundary is the Python call level (SURVEY.md
 * section 8b).  Each entry point below names the reference call it replaces
 * (paths relative to the reference tree; notebooks cited by raw .ipynb JSON line).
 *
 * Conventions
 *   * Plain C, no torch / HIP types in signatures.  `stream` is a hipStream_t
 *     passed as void* (NULL = the default stream).
 *   * `*_dev` entry points take DEVICE pointers, enqueue on `stream` and return
 *     without synchronising (safe to capture in a hipGraph).  The un-suffixed
 *     twins take HOST pointers, stage through the context's device workspace and
 *     synchronise before returning.
 *   * Every function returns a tda_status (0 = ok); nothing throws or aborts.
 *     Per-window problems are reported in the `status` arrays (bit flags below),
 *     never silently.
 *   * The library keeps no pointer after a call returns and never writes inputs.
 *   * Persistence diagrams are rows of (birth, death) float64 holding float32-exact
 *     values (what ripser returns), +inf for essential classes.
 *       H0: finite rows in ascending death order, then one (0,+inf) per component.
 *       H1: rows in descending birth order (ripser's column order).
 */
#ifndef TDAEEG_H
#define TDAEEG_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tda_ctx tda_ctx;

typedef enum {
    TDA_OK = 0,
    TDA_ERR_INVALID = 1,      /* bad argument (null pointer, size out of range) */
    TDA_ERR_HIP = 2,          /* a HIP runtime call failed; see tda_last_error */
    TDA_ERR_UNSUPPORTED = 3,  /* size outside what the kernels implement        */
    TDA_ERR_NOMEM = 4
} tda_status;

/* per-window status bits written by the kernels */
#define TDA_WIN_OK            0
#define TDA_WIN_H1_TRUNCATED  1   /* more H1 rows than h1_cap; count holds the true number    */
#define TDA_WIN_CLASS_OVERFLOW 2  /* more simultaneously alive H1 classes than the kernel's
                                     class capacity (tda_set_class_words); diagrams invalid   */
#define TDA_WIN_DEGENERATE    4   /* point cloud with < 3 points: diagrams are [[0,0]],[[0,0]]
                                     exactly as scripts/utils.py:125-126                      */
#define TDA_WIN_NOT_CONVERGED 8   /* assignment solver hit its iteration bound (-> NaN)        */
#define TDA_WIN_TOO_LARGE     16  /* more than TDA_MAX_POINTS points (tau < 1?): no diagram    */

#define TDA_N_FEATURES 11         /* scripts/utils.py:166-177 key order */
#define TDA_MAX_POINTS 128        /* vertices per Rips complex (reference needs <= 124)        */
#define TDA_MAX_DIM    4          /* Takens embedding dimension (reference uses 3)             */

int  tda_version(void);

/* Context: owns the HIP device binding, a stream-ordered workspace and the error
 * string.  One per process and GPU (the reference's joblib workers, v2:569-572,
 * become one process per GPU). */
tda_status tda_ctx_create(int device_id, tda_ctx** out);
void       tda_ctx_destroy(tda_ctx* ctx);
/* Copies the last error message of this context (or of the failed create when
 * ctx == NULL) into buf; returns its length. */
size_t     tda_last_error(const tda_ctx* ctx, char* buf, size_t cap);
/* First-pass capacity for simultaneously alive H1 classes.  words_dm in {0,1,2,4}: 64 * words bits for distance-matrix
 * input; 0 = 32 bits, honoured by the fused EEG window kernel only (tda_rips_dm_batch treats it as 1).  words_cloud in
 * {1,2}: 32 or 64 bits for point clouds.  Defaults: 2 and 1.  Windows that need more are redone by the widening passes
 * (tda_set_retry_policy), so the setting changes speed, never results. */
tda_status tda_set_class_words(tda_ctx* ctx, int words_dm, int words_cloud);
/* How the Rips entry points treat windows that run out of class bits (TDA_WIN_CLASS_OVERFLOW).
 * AUTO (default): every call launches its widening passes after the first pass; they redo only the flagged
 *   windows, but each is a launch that needs (large) CU resources even when nothing is flagged.
 * FIRST_PASS: only the first pass is launched; flagged windows keep the status bit and invalid rows.  A
 *   streaming caller checks the statuses once they have reached the host and, if any is set, calls the same
 *   entry point again on the same buffers under RETRY_ONLY (then recomputes what depends on the diagrams).
 * RETRY_ONLY: only the widening passes (and the row ordering) are launched.
 * ONE_STEP: the first pass and ONE widening pass (the next rung of the ladder: 128 bits for matrices, 64 for
 *   clouds), which catches nearly every flagged window and still fits beside the other kernels of a busy GPU; the
 *   wide rungs (up to 95 KB of LDS and 256 VGPRs per workgroup, which wait for a nearly empty CU even when they have
 *   nothing to redo) are left to a later RETRY_ONLY call for the batches whose statuses still carry the bit.
 * The ladders end in a pass that keeps the class vectors in HBM and has no capacity limit (8,192 classes cover every
 * complex on 128 points): under AUTO and RETRY_ONLY no window keeps TDA_WIN_CLASS_OVERFLOW, as ripser never refuses an
 * input.  LAST_RUNG launches only that pass (on the flagged windows). */
#define TDA_RETRY_AUTO       0
#define TDA_RETRY_FIRST_PASS 1
#define TDA_RETRY_ONLY       2
#define TDA_RETRY_ONE_STEP   3
#define TDA_RETRY_LAST_RUNG  4   /* only the last rung (class vectors in HBM) on the flagged windows: for tests */
tda_status tda_set_retry_policy(tda_ctx* ctx, int policy);
/* Optional accounting of the widening passes: dev_counters = DEVICE u64[4] (or NULL to stop).  Every window a
 * widening pass redoes adds one to [0] (distance-matrix input, tda_rips_dm_batch) or [1] (point clouds,
 * tda_takens_rips_batch / tda_cloud_rips_batch); a window that climbs two rungs of the ladder counts twice; [2] counts
 * the windows redone by the last rung (class vectors in HBM, no capacity limit).
 * The reference has no counterpart (ripser's columns grow on the heap); bench.py reports it as windows_repaired. */
tda_status tda_set_retry_counter(tda_ctx* ctx, void* dev_counters);

/* ---- corr -> distance ------------------------------------------------------
 * replaces compute_correlation_matrix + correlation_to_distance(method="euclidean")
 * (notebooks/2_graph_construction.ipynb:86-122) and the per-window loop of
 * process_file_graphs (nb2:198-207).
 * win  : (n_win, n_ch, n_t) float64, C order   (preprocessed/<cond>/<rec>/<band>.npy)
 * dist : (n_win, n_ch, n_ch) float64           (<band>_distances.npy)
 * corr : same shape or NULL                    (<band>_correlations.npy)            */
tda_status tda_corr_dist_batch_dev(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t,
                                   double* dist, double* corr, void* stream);
tda_status tda_corr_dist_batch(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t,
                               double* dist, double* corr);

/* Sliding windows fused in: create_sliding_windows (notebooks/1_preprocesamiento.ipynb:314-381;
 * window w = samples [w*step, w*step+win_len), n_win = (n_samples-win_len)/step+1) followed by the
 * per-window corr->dist above, reading the overlapping windows straight from ONE band-passed
 * recording sig (n_ch, n_samples) float64 -- the (n_win, n_ch, win_len) stack (4x the bytes at 75 %
 * overlap) is never materialised.  dist/corr: (n_win, n_ch, n_ch); n_win (out, nullable).      */
tda_status tda_corr_dist_sliding_dev(tda_ctx* ctx, const double* sig, int n_ch, int n_samples,
                                     int win_len, int step, double* dist, double* corr, int* n_win,
                                     void* stream);
tda_status tda_corr_dist_sliding(tda_ctx* ctx, const double* sig, int n_ch, int n_samples,
                                 int win_len, int step, double* dist, double* corr, int* n_win);

/* correlation_to_distance (nb2:100-122) alone, on stored correlation matrices.
 * method: 0 "euclidean" (the only one the reference calls, nb2:227,304,312), 1 "abs",
 *         2 "standard", 3 "sqrt" (nb2:109-116).  corr, dist: (n_win, n, n) float64.          */
tda_status tda_corr_to_dist_batch_dev(tda_ctx* ctx, const double* corr, int n_win, int n, int method,
                                      double* dist, void* stream);
tda_status tda_corr_to_dist_batch(tda_ctx* ctx, const double* corr, int n_win, int n, int method,
                                  double* dist);

/* ---- Vietoris-Rips H0/H1 from distance matrices -----------------------------
 * replaces compute_eeg_persistence (scripts/utils.py:135-141) ==
 * compute_persistence_diagram (scripts/tda_eeg_classification_v2.py:143-176):
 * ripser(dm, maxdim=1, thresh, distance_matrix=True)["dgms"].
 * dm        : (n_win, n, n) float64, n <= TDA_MAX_POINTS
 * symmetrise: 1 = apply (D+D^T)/2, diag 0, max(.,0) first (utils.py:137-139);
 *             0 = use entries (i,j), i<j, as ripser.py does on a raw matrix
 * h0        : (n_win, h0_cap, 2) float64, h0_cap >= n;   h0_cnt: (n_win) int32
 * h1        : (n_win, h1_cap, 2) float64;                h1_cnt: (n_win) int32
 * status    : (n_win) int32 bit flags (TDA_WIN_*)                                    */
tda_status tda_rips_dm_batch_dev(tda_ctx* ctx, const double* dm, int n_win, int n, double thresh,
                                 int symmetrise, double* h0, int h0_cap, int* h0_cnt,
                                 double* h1, int h1_cap, int* h1_cnt, int* status, void* stream);
tda_status tda_rips_dm_batch(tda_ctx* ctx, const double* dm, int n_win, int n, double thresh,
                             int symmetrise, double* h0, int h0_cap, int* h0_cnt,
                             double* h1, int h1_cap, int* h1_cnt, int* status);

/* ---- fused EEG window: samples -> correlation -> distance -> Rips H0/H1 in ONE launch -------------
 * replaces, per window, compute_correlation_matrix + correlation_to_distance (nb2:86-122; the loop of
 * process_file_graphs, nb2:198-207) followed by compute_eeg_persistence (scripts/utils.py:135-141): the distance
 * matrix stays in LDS, 95.1 KB of HBM traffic per window instead of 129.4.  Same arithmetic, operation for
 * operation, as tda_corr_dist_batch_dev + tda_rips_dm_batch_dev(symmetrise = 1): identical diagrams.
 * win: (n_win, n_ch, n_t) float64, 33 <= n_ch <= 48 (the reference has 47), n_t <= 256 (250).
 * dist / corr (nullable; corr needs dist): the matrices as tda_corr_dist_batch writes them, for callers that also
 * want the graphs/<cond>/<rec>/<band>_distances.npy hand-off file.  Device pointers only. */
tda_status tda_eeg_window_batch_dev(tda_ctx* ctx, const double* win, int n_win, int n_ch, int n_t, double thresh,
                                    double* dist, double* corr, double* h0, int h0_cap, int* h0_cnt,
                                    double* h1, int h1_cap, int* h1_cnt, int* status, void* stream);

/* The same kernel on windows read IN PLACE from band-passed recordings of equal length, sig: (n_rec, n_ch,
 * n_samples) float64 -- create_sliding_windows (notebooks/1_preprocesamiento.ipynb:314-381: window k of a recording =
 * samples [k*step, k*step + win_len), n_win_per_rec = (n_samples - win_len) / step + 1) + process_file_graphs
 * (nb2:198-207) + compute_eeg_persistence (utils.py:135-141) without ever materialising the (n_win, n_ch, win_len)
 * stack (4x the bytes at 75 % overlap; 43 GB for the corpus) or the matrices.  sel (nullable, n_sel entries): the
 * windows to process, as r * n_win_per_rec + k -- the drivers' window selection (v2:394-398, cmp:77-80); outputs
 * have n_sel rows then, otherwise n_rec * n_win_per_rec (recording-major).  n_win_per_rec (out, nullable, host). */
tda_status tda_eeg_window_sliding_dev(tda_ctx* ctx, const double* sig, int n_rec, int n_ch, int n_samples,
                                      int win_len, int step, const int* sel, int n_sel, double thresh,
                                      double* dist, double* corr, double* h0, int h0_cap, int* h0_cnt,
                                      double* h1, int h1_cap, int* h1_cnt, int* status, int* n_win_per_rec,
                                      void* stream);

/* ---- Takens embedding + Rips (audio branch) ---------------------------------
 * replaces takens_embedding (scripts/utils.py:107-116) followed by
 * compute_audio_persistence (utils.py:123-132): per-column min-max to [0,1],
 * ripser(pc_norm, maxdim=1, thresh) whose point-cloud path is
 * sklearn.metrics.pairwise_distances -> float32.
 * win  : (n_win, n_t) float64 audio windows;  tau: (n_win) int32 delays
 * n_points (out, nullable): points per window, P = ceil((n_t-(dim-1)tau)/subsample)
 * Windows with P < 3 get [[0,0]],[[0,0]] and TDA_WIN_DEGENERATE (utils.py:125-126).  */
tda_status tda_takens_rips_batch_dev(tda_ctx* ctx, const double* win, const int* tau, int n_win,
                                     int n_t, int dim, int subsample, double thresh,
                                     double* h0, int h0_cap, int* h0_cnt,
                                     double* h1, int h1_cap, int* h1_cnt,
                                     int* n_points, int* status, void* stream);
tda_status tda_takens_rips_batch(tda_ctx* ctx, const double* win, const int* tau, int n_win,
                                 int n_t, int dim, int subsample, double thresh,
                                 double* h0, int h0_cap, int* h0_cnt,
                                 double* h1, int h1_cap, int* h1_cnt,
                                 int* n_points, int* status);

/* ---- Rips from explicit point clouds ----------------------------------------
 * replaces compute_audio_persistence(point_cloud) (utils.py:123-132) when the caller
 * already holds the (P, dim) cloud.  pc: (n_win, p_cap, dim) float64; n_pts: (n_win). */
tda_status tda_cloud_rips_batch_dev(tda_ctx* ctx, const double* pc, const int* n_pts, int n_win,
                                    int p_cap, int dim, int normalise, double thresh,
                                    double* h0, int h0_cap, int* h0_cnt,
                                    double* h1, int h1_cap, int* h1_cnt, int* status, void* stream);
tda_status tda_cloud_rips_batch(tda_ctx* ctx, const double* pc, const int* n_pts, int n_win,
                                int p_cap, int dim, int normalise, double thresh,
                                double* h0, int h0_cap, int* h0_cnt,
                                double* h1, int h1_cap, int* h1_cnt, int* status);

/* ---- zero-phase IIR filtering (front ends) ----------------------------
 * tda_sosfiltfilt replaces scipy.signal.sosfiltfilt(sos, x) as apply_bandpass_filter calls it per
 * EEG channel (notebooks/1_preprocesamiento.ipynb:236-263); tda_filtfilt replaces
 * scipy.signal.filtfilt(b, a, s) of bandpass_filter (scripts/utils.py:66-74).  Same algorithm as
 * scipy (odd extension by `edge`, forward from zi*x[0], backward from zi*y[-1], trim), same
 * operation order: bit-identical float64 results.  Filter design stays on the host:
 *   sos (n_sections,6) from scipy.signal.butter(..., output="sos"), zi (n_sections,2) from sosfilt_zi,
 *   edge = 3*(2*n_sections+1 - min(#b2==0, #a2==0));   b, a (ntaps), zi (ntaps-1) from lfilter_zi,
 *   edge = 3*ntaps.
 * x, y: (n_sig, n_samples) float64; work (device, *_dev only): (n_sig, n_samples + 2*edge).
 * sos / zi / b / a are HOST pointers in both forms (they travel as kernel arguments).           */
tda_status tda_sosfiltfilt_dev(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* sos,
                               const double* zi, int n_sections, int edge, double* y, double* work,
                               void* stream);
tda_status tda_sosfiltfilt(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* sos,
                           const double* zi, int n_sections, int edge, double* y);
tda_status tda_filtfilt_dev(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* b,
                            const double* a, const double* zi, int ntaps, int edge, double* y, double* work,
                            void* stream);
tda_status tda_filtfilt(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* b,
                        const double* a, const double* zi, int ntaps, int edge, double* y);

/* Filter BANKS: the same n_sig signals through n_filters filters of equal structure in ONE launch -- the five frequency
 * bands of apply_bandpass_filter (nb1:236-263, one call per band in preprocess_file nb1:388-494) and of bandpass_filter
 * (utils.py:66-74, one call per band in process_recording, cmp:63-64).  sos (n_filters, n_sections, 6), zi (n_filters,
 * n_sections, 2) / b, a (n_filters, ntaps), zi (n_filters, ntaps-1): HOST pointers; y (n_filters, n_sig, n_samples),
 * work (n_filters, n_sig, n_samples + 2*edge): device.  n_filters <= 5.  Bit-identical to the single-filter calls. */
tda_status tda_sosfiltfilt_bank_dev(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* sos,
                                    const double* zi, int n_filters, int n_sections, int edge, double* y, double* work,
                                    void* stream);
tda_status tda_filtfilt_bank_dev(tda_ctx* ctx, const double* x, int n_sig, int n_samples, const double* b,
                                 const double* a, const double* zi, int n_filters, int ntaps, int edge, double* y,
                                 double* work, void* stream);

/* Audio front end.  tda_upfirdn replaces scipy.signal.resample_poly(audio, 250, 44100) (scripts/utils.py:77-79):
 * h (len_h, host-designed exactly as scipy does, incl. its zero padding and the factor `up`),
 *   y[j] = sum_i x[i] * h[(j + n_pre_remove)*down - i*up],  j < n_out.
 * tda_hilbert_envelope replaces np.abs(scipy.signal.hilbert(s)) (utils.py:58-59):
 *   env[n] = sqrt(x[n]^2 + (sum_m x[m] g[(n-m) mod N])^2),  g = imag(ifft(h_hilbert)) tabulated by the host.
 * float64, agreement with scipy to rounding (1e-12 relative), not bit-identical.  *_dev: all pointers device. */
tda_status tda_upfirdn_dev(tda_ctx* ctx, const double* x, long long n_in, const double* h, int len_h, int up,
                           int down, long long n_pre_remove, long long n_out, double* y, void* stream);
tda_status tda_upfirdn(tda_ctx* ctx, const double* x, long long n_in, const double* h, int len_h, int up,
                       int down, long long n_pre_remove, long long n_out, double* y);
tda_status tda_hilbert_envelope_dev(tda_ctx* ctx, const double* x, int n, const double* g, double* env,
                                    void* stream);
tda_status tda_hilbert_envelope(tda_ctx* ctx, const double* x, int n, const double* g, double* env);

/* ---- delay from the first zero crossing of the autocorrelation ---------------
 * replaces compute_tau (scripts/utils.py:92-104). max_lag < 0 = None (len/4).         */
tda_status tda_tau_batch_dev(tda_ctx* ctx, const double* win, int n_win, int n_t, int max_lag,
                             int* tau, void* stream);
tda_status tda_tau_batch(tda_ctx* ctx, const double* win, int n_win, int n_t, int max_lag, int* tau);
/* The driver's use of it (scripts/tda_eeg_audio_comparison.py:83, scripts/matched_vs_mismatched.py:56): one tau per
 * (recording, band) group, from the FIRST window of the group (window seg_off[g] of win).  tau_seg: (n_seg);
 * tau_win (nullable): (n_total) receives the group's value for each of its windows, the per-window array
 * tda_takens_rips_batch takes.  Device pointers only (a stage of the batched driver). */
tda_status tda_tau_segments_dev(tda_ctx* ctx, const double* win, const int* seg_off, int n_seg, int n_t,
                                int max_lag, int* tau_seg, int* tau_win, void* stream);

/* ---- 11 scalar features per diagram ------------------------------------------
 * replaces extract_features (scripts/utils.py:144-177) ==
 * extract_persistence_features (tda_eeg_classification_v2.py:179-250).
 * dgm: (n_dgm, cap, 2) float64; cnt: (n_dgm); feat: (n_dgm, 11) float64.             */
tda_status tda_features_batch_dev(tda_ctx* ctx, const double* dgm, const int* cnt, int n_dgm,
                                  int cap, double* feat, void* stream);
tda_status tda_features_batch(tda_ctx* ctx, const double* dgm, const int* cnt, int n_dgm,
                              int cap, double* feat);

/* ---- finishing pass over the diagrams of a batch ---------------------------------
 * Up to four diagram sets in ONE launch (one wavefront per diagram): for sets with order != 0 the H1 rows are put
 * into ripser's order in place, and where feat != NULL the 11 scalars of extract_features are written -- what
 * tda_features_batch_dev does for one set.  The driver of a whole step runs the Rips entry points under
 * TDA_ORDER_DEFERRED and finishes the EEG H0 / EEG H1 / audio H1 diagrams of the batch with one call
 * (scripts/tda_eeg_audio_comparison.py:92-99; scripts/tda_eeg_classification_v2.py:410-416).  `sets` is a host array
 * of descriptors; the pointers inside are device pointers. */
typedef struct {
    double* rows;      /* (n_dgm, cap, 2) float64 */
    const int* cnt;    /* (n_dgm) */
    int cap;
    int order;         /* 1: rows are H1 rows in emission order -> descending birth (ripser's order) */
    double* feat;      /* (n_dgm, 11) float64 or NULL */
} tda_diagram_set;
tda_status tda_diagram_finish_dev(tda_ctx* ctx, const tda_diagram_set* sets, int n_sets, int n_dgm, void* stream);
/* IN_CALL (default): every Rips entry point returns H1 rows in ripser's order (it launches the finishing pass for
 * its own diagrams).  DEFERRED: rows stay in emission order until the caller's tda_diagram_finish_dev. */
#define TDA_ORDER_IN_CALL  0
#define TDA_ORDER_DEFERRED 1
tda_status tda_set_h1_order(tda_ctx* ctx, int policy);

/* ---- per-recording aggregation -------------------------------------------------
 * replaces the mean/std over windows of process_file_features
 * (tda_eeg_classification_v2.py:429-436).
 * feat_h0, feat_h1 : (n_total, 11) float64 features of the used windows, grouped
 * seg_off          : (n_seg+1) int32 offsets of each (recording, band) group
 * out              : (n_seg, 44) float64 in the column order of features/feature_names.txt
 *                    within one band: per feature {h0 mean, h0 std, h1 mean, h1 std}.  */
tda_status tda_aggregate_batch_dev(tda_ctx* ctx, const double* feat_h0, const double* feat_h1,
                                   const int* seg_off, int n_seg, double* out, void* stream);
tda_status tda_aggregate_batch(tda_ctx* ctx, const double* feat_h0, const double* feat_h1,
                               const int* seg_off, int n_seg, int n_total, double* out);

/* ---- np.nanmean over the windows of each (recording, band) group ----------------
 * replaces np.nanmean(wass_h0) / np.nanmean(vals) (scripts/tda_eeg_audio_comparison.py:117-118,
 * scripts/matched_vs_mismatched.py:95).  x: (n_total) float64; seg_off: (n_seg+1) int32.
 * Empty or all-NaN groups give NaN.                                                    */
tda_status tda_segment_nanmean_dev(tda_ctx* ctx, const double* x, const int* seg_off, int n_seg,
                                   double* out, void* stream);
tda_status tda_segment_nanmean(tda_ctx* ctx, const double* x, const int* seg_off, int n_seg,
                               int n_total, double* out);

/* ---- one result row per (recording, band) group ---------------------------------------
 * out: (n_seg, 48) float64 = [ nanmean of w_h0 (cmp:117), nanmean of w_h1 (cmp:118), tau (cmp:83), number of
 * windows, the 44 values of tda_aggregate_batch (v2:429-436) ] -- tda_segment_nanmean x 2 + tda_aggregate_batch
 * + the row assembly in ONE launch; the rows are what the GPUs of a node exchange.  Device pointers only.
 * status_a / status_b / seg_flags (all nullable): per-window status arrays of the two Rips calls and an
 * (n_seg) int32 output that receives, per group, the OR of their status words without TDA_WIN_DEGENERATE (a result,
 * not a condition) -- what a caller running under TDA_RETRY_FIRST_PASS / ONE_STEP copies to the host: bit
 * TDA_WIN_CLASS_OVERFLOW asks for the rest of the ladder, any other bit means rows the reference would not give. */
tda_status tda_recording_rows_dev(tda_ctx* ctx, const double* w_h0, const double* w_h1, const int* tau_seg,
                                  const double* feat_h0, const double* feat_h1, const int* seg_off,
                                  int n_seg, double* out, const int* status_a, const int* status_b,
                                  int* seg_flags, void* stream);

/* ---- Spearman correlation of feature time series ------------------------------------
 * replaces the spearmanr(a_ts, e_ts) loop of process_recording
 * (scripts/tda_eeg_audio_comparison.py:104-114): per (recording, band) group and per selected
 * feature column, the Pearson correlation of the average ranks of the audio and EEG series;
 * r = 0 when the group has < 5 windows or a series has np.std <= 1e-10 (the reference's rule,
 * which then reports p = 1).  x, y: (n_total, ld) float64; cols: (n_cols) column indices;
 * r: (n_seg, n_cols).  The p-value is a function of (r, n) only (Student t) and is left to the host. */
tda_status tda_spearman_batch_dev(tda_ctx* ctx, const double* x, const double* y, int ld, const int* cols,
                                  int n_cols, const int* seg_off, int n_seg, double* r, void* stream);
tda_status tda_spearman_batch(tda_ctx* ctx, const double* x, const double* y, int n_total, int ld,
                              const int* cols, int n_cols, const int* seg_off, int n_seg, double* r);

/* ---- Wasserstein distance between diagrams ------------------------------------
 * replaces safe_wasserstein (scripts/utils.py:180-191) -> persim.wasserstein
 * (order 1, Euclidean ground metric, diagonal cost (d-b)/sqrt 2).
 * dgm_a: (n_a, cap_a, 2), cnt_a: (n_a);  dgm_b likewise.
 * idx_a, idx_b: (n_pairs) int32 diagram indices to pair (NULL = identity).
 * Rows with a non-finite entry are ignored and an empty diagram becomes {(0,0)}
 * (utils.py:182-187).  out: (n_pairs) float64, NaN where status != 0.                 */
tda_status tda_wasserstein_batch_dev(tda_ctx* ctx, const double* dgm_a, const int* cnt_a, int cap_a,
                                     const double* dgm_b, const int* cnt_b, int cap_b,
                                     const int* idx_a, const int* idx_b, int n_pairs,
                                     double* out, int* status, void* stream);
tda_status tda_wasserstein_batch(tda_ctx* ctx, const double* dgm_a, const int* cnt_a, int n_a, int cap_a,
                                 const double* dgm_b, const int* cnt_b, int n_b, int cap_b,
                                 const int* idx_a, const int* idx_b, int n_pairs,
                                 double* out, int* status);

/* ---- timing helper ------------------------------------------------------------
 * HIP-event timing on the stream the kernels are launched on (bench.py roofline). */
tda_status tda_event_create(tda_ctx* ctx, void** ev);
tda_status tda_event_record(tda_ctx* ctx, void* ev, void* stream);
tda_status tda_event_elapsed_ms(tda_ctx* ctx, void* ev_start, void* ev_stop, float* ms); /* syncs on stop */
tda_status tda_event_destroy(tda_ctx* ctx, void* ev);
tda_status tda_stream_sync(tda_ctx* ctx, void* stream);
/* Arms a one-shot probe (one slot per `which`; TDA_PROBE_NONE clears all): the NEXT launch of the first-pass kernel of `which` made through this
 * context records ev_start right before and ev_stop right after that ONE kernel on its stream
 * (the retry passes and the row-ordering kernel of the same call are outside the bracket).  This is
 * the per-kernel duration bench.py's roofline uses; rocprofv3 --kernel-trace reports the same kernel.
 * dev_span (optional, TDA_PROBE_RIPS_CLOUD only): device u64[4] preset to 0.  Workgroup 0 (dispatched
 * first) stores the 100 MHz wall-clock time of its start in [0]; finished workgroups are counted in [1];
 * the last one adds (its end - that start) -- the interval a kernel trace shows, i.e. without the time
 * the grid waits for CU slots behind other in-flight batches -- to [2], counts the launch in [3] and
 * resets [1].  A launch captured into a HIP graph with the probe armed therefore measures every replay;
 * average duration = [2] / [3] / 100 MHz.  The events may be NULL when dev_span is given. */
#define TDA_PROBE_NONE       0
#define TDA_PROBE_RIPS_CLOUD 1   /* rips_cloud_kernel, first pass (tda_takens_rips_batch / tda_cloud_rips_batch) */
#define TDA_PROBE_RIPS_DM    2   /* rips_dm_kernel, first pass (tda_rips_dm_batch) */
#define TDA_PROBE_CORR_DIST  3   /* corr_dist_kernel (tda_corr_dist_batch / _sliding) */
tda_status tda_set_kernel_probe(tda_ctx* ctx, int which, void* ev_start, void* ev_stop, void* dev_span);

#ifdef __cplusplus
}
#endif
#endif /* TDAEEG_H */

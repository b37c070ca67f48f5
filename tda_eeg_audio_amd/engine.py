"""
engine.py -- batched numpy / torch-tensor front end of libtdaeeg.so.

Every function here is a thin marshalling layer over one C-ABI entry point of
include/tdaeeg.h; all arithmetic happens in the HIP kernels.  There is no CPU path.

  host arrays  : corr_dist_batch, rips_dm_batch, takens_rips_batch, cloud_rips_batch,
                 tau_batch, features_batch, aggregate_batch, wasserstein_batch
  device tensors (torch, already resident in HBM, launched on torch's current stream):
                 the ``*_dev`` twins -- used by bench.py and the multi-GPU driver.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import f64, i32, ptr, get_ctx

MAX_EDGE_LENGTH = 2.0      # scripts/utils.py:25
DEFAULT_H1_CAP = 256


def _diagrams(rows, cnt, cap):
    """(n, cap, 2) rows + counts -> list of (k,2) float64 arrays."""
    return [rows[i, :min(int(cnt[i]), cap)].copy() for i in range(rows.shape[0])]


# ------------------------------------------------------------------ host-array API
def corr_dist_batch(windows, want_corr=True, ctx=None):
    ctx = ctx or get_ctx()
    w = f64(windows)
    n_win, n_ch, n_t = w.shape
    dist = np.empty((n_win, n_ch, n_ch))
    corr = np.empty((n_win, n_ch, n_ch)) if want_corr else None
    ctx.check(ctx.lib.tda_corr_dist_batch(ctx.h, ptr(w), n_win, n_ch, n_t, ptr(dist), ptr(corr)))
    return (corr, dist) if want_corr else dist


def corr_dist_sliding(signal, win_len=250, step=62, want_corr=False, ctx=None):
    """(n_ch, n_samples) band-passed recording -> distance matrices of its sliding windows
    (nb1:314-381 + nb2:198-207 fused)."""
    ctx = ctx or get_ctx()
    s = f64(signal)
    n_ch, n_s = s.shape
    n_win = (n_s - win_len) // step + 1 if n_s >= win_len else 0
    dist = np.empty((n_win, n_ch, n_ch))
    corr = np.empty((n_win, n_ch, n_ch)) if want_corr else None
    ctx.check(ctx.lib.tda_corr_dist_sliding(ctx.h, ptr(s), n_ch, n_s, win_len, step, ptr(dist), ptr(corr), None))
    return (corr, dist) if want_corr else dist


def corr_dist_sliding_dev(sig_t, win_len=250, step=62, dist_t=None, corr_t=None, ctx=None):
    import torch
    ctx = ctx or get_ctx()
    assert sig_t.is_cuda and sig_t.dtype == torch.float64 and sig_t.is_contiguous()
    n_ch, n_s = sig_t.shape
    n_win = (n_s - win_len) // step + 1 if n_s >= win_len else 0
    if dist_t is None:
        dist_t = torch.empty((n_win, n_ch, n_ch), dtype=torch.float64, device=sig_t.device)
    ctx.check(ctx.lib.tda_corr_dist_sliding_dev(ctx.h, _tp(sig_t), n_ch, n_s, win_len, step, _tp(dist_t), _tp(corr_t),
                                                None, _stream()))
    return dist_t


DIST_METHODS = {"euclidean": 0, "abs": 1, "standard": 2, "sqrt": 3}     # nb2:107-116


def corr_to_dist_batch(corr, method="euclidean", ctx=None):
    if method not in DIST_METHODS:
        raise ValueError(f"Unknown method: {method}")                      # nb2:117
    ctx = ctx or get_ctx()
    c = f64(corr)
    n_win, n, _ = c.shape
    dist = np.empty_like(c)
    ctx.check(ctx.lib.tda_corr_to_dist_batch(ctx.h, ptr(c), n_win, n, DIST_METHODS[method], ptr(dist)))
    return dist


def rips_dm_batch(dms, thresh=MAX_EDGE_LENGTH, symmetrise=True, h1_cap=DEFAULT_H1_CAP, ctx=None, raw=False):
    ctx = ctx or get_ctx()
    d = f64(dms)
    n_win, n, n2 = d.shape
    assert n == n2, "distance matrices must be square"   # the only thing ripser itself rejects
    h0 = np.empty((n_win, n, 2)); h1 = np.empty((n_win, h1_cap, 2))
    c0 = np.empty(n_win, np.int32); c1 = np.empty(n_win, np.int32); st = np.empty(n_win, np.int32)
    ctx.check(ctx.lib.tda_rips_dm_batch(ctx.h, ptr(d), n_win, n, float(thresh), int(bool(symmetrise)),
                                        ptr(h0), n, ptr(c0), ptr(h1), h1_cap, ptr(c1), ptr(st)))
    if raw:
        return h0, c0, h1, c1, st
    return _diagrams(h0, c0, n), _diagrams(h1, c1, h1_cap), st


def takens_rips_batch(windows, taus, dim=3, subsample=2, thresh=MAX_EDGE_LENGTH, h1_cap=DEFAULT_H1_CAP,
                      ctx=None, raw=False):
    ctx = ctx or get_ctx()
    w = f64(windows)
    n_win, n_t = w.shape
    tau = i32(np.broadcast_to(np.asarray(taus), (n_win,)))
    h0_cap = _lib.MAX_POINTS
    h0 = np.empty((n_win, h0_cap, 2)); h1 = np.empty((n_win, h1_cap, 2))
    c0 = np.empty(n_win, np.int32); c1 = np.empty(n_win, np.int32)
    npts = np.empty(n_win, np.int32); st = np.empty(n_win, np.int32)
    ctx.check(ctx.lib.tda_takens_rips_batch(ctx.h, ptr(w), ptr(tau), n_win, n_t, dim, subsample, float(thresh),
                                            ptr(h0), h0_cap, ptr(c0), ptr(h1), h1_cap, ptr(c1), ptr(npts), ptr(st)))
    if raw:
        return h0, c0, h1, c1, npts, st
    return _diagrams(h0, c0, h0_cap), _diagrams(h1, c1, h1_cap), npts, st


def cloud_rips_batch(clouds, n_pts=None, normalise=True, thresh=MAX_EDGE_LENGTH, h1_cap=DEFAULT_H1_CAP,
                     ctx=None, raw=False):
    ctx = ctx or get_ctx()
    pc = f64(clouds)
    n_win, p_cap, dim = pc.shape
    n_pts = i32(np.full(n_win, p_cap) if n_pts is None else n_pts)
    h0_cap = max(p_cap, 3)
    h0 = np.empty((n_win, h0_cap, 2)); h1 = np.empty((n_win, h1_cap, 2))
    c0 = np.empty(n_win, np.int32); c1 = np.empty(n_win, np.int32); st = np.empty(n_win, np.int32)
    ctx.check(ctx.lib.tda_cloud_rips_batch(ctx.h, ptr(pc), ptr(n_pts), n_win, p_cap, dim, int(bool(normalise)),
                                           float(thresh), ptr(h0), h0_cap, ptr(c0), ptr(h1), h1_cap, ptr(c1),
                                           ptr(st)))
    if raw:
        return h0, c0, h1, c1, st
    return _diagrams(h0, c0, h0_cap), _diagrams(h1, c1, h1_cap), st


def tau_batch(windows, max_lag=None, ctx=None):
    ctx = ctx or get_ctx()
    w = f64(windows)
    n_win, n_t = w.shape
    tau = np.empty(n_win, np.int32)
    ctx.check(ctx.lib.tda_tau_batch(ctx.h, ptr(w), n_win, n_t, -1 if max_lag is None else int(max_lag), ptr(tau)))
    return tau


def pack_diagrams(dgms, cap=None):
    """list of (k,2) arrays -> (n, cap, 2) float64 + counts."""
    arrs = [np.asarray(d, dtype=np.float64).reshape(-1, 2) if np.asarray(d).ndim == 2 and np.asarray(d).size
            else np.zeros((0, 2)) for d in dgms]
    cap = cap or max(1, max((a.shape[0] for a in arrs), default=1))
    rows = np.zeros((len(arrs), cap, 2))
    cnt = np.zeros(len(arrs), np.int32)
    for i, a in enumerate(arrs):
        assert a.shape[0] <= cap
        rows[i, :a.shape[0]] = a
        cnt[i] = a.shape[0]
    return rows, cnt


def features_batch(rows, cnt, ctx=None):
    ctx = ctx or get_ctx()
    rows = f64(rows); cnt = i32(cnt)
    n, cap, _ = rows.shape
    feat = np.empty((n, _lib.N_FEATURES))
    ctx.check(ctx.lib.tda_features_batch(ctx.h, ptr(rows), ptr(cnt), n, cap, ptr(feat)))
    return feat


def aggregate_batch(feat_h0, feat_h1, seg_off, ctx=None):
    ctx = ctx or get_ctx()
    f0 = f64(feat_h0); f1 = f64(feat_h1); off = i32(seg_off)
    n_seg = len(off) - 1
    out = np.empty((n_seg, 4 * _lib.N_FEATURES))
    ctx.check(ctx.lib.tda_aggregate_batch(ctx.h, ptr(f0), ptr(f1), ptr(off), n_seg, f0.shape[0], ptr(out)))
    return out


def segment_nanmean(x, seg_off, ctx=None):
    ctx = ctx or get_ctx()
    x = f64(x); off = i32(seg_off)
    out = np.empty(len(off) - 1)
    ctx.check(ctx.lib.tda_segment_nanmean(ctx.h, ptr(x), ptr(off), len(off) - 1, x.shape[0], ptr(out)))
    return out


SPEARMAN_FEATURES = ["mean_persistence", "total_persistence", "persistence_entropy", "max_persistence", "n_features"]
SPEARMAN_COLS = [6, 9, 10, 8, 0]          # their columns in the 11-feature vector (cmp:106-107)


def spearman_batch(feat_a, feat_b, seg_off, cols=SPEARMAN_COLS, ctx=None):
    """r (n_seg, len(cols)) and the two-sided p-value (scipy's Student-t formula on the host)."""
    ctx = ctx or get_ctx()
    fa = f64(feat_a); fb = f64(feat_b); off = i32(seg_off); cc = i32(cols)
    n_seg = len(off) - 1
    r = np.empty((n_seg, len(cc)))
    ctx.check(ctx.lib.tda_spearman_batch(ctx.h, ptr(fa), ptr(fb), fa.shape[0], fa.shape[1], ptr(cc), len(cc), ptr(off),
                                         n_seg, ptr(r)))
    from scipy import stats
    n = np.diff(off).astype(float)[:, None]
    dof = n - 2
    with np.errstate(divide="ignore", invalid="ignore"):
        t = r * np.sqrt((dof / ((r + 1.0) * (1.0 - r))).clip(0))
        p = 2 * stats.t.sf(np.abs(t), dof)
    # the reference reports r = 0, p = 1 for short or constant series (cmp:113-114)
    short = (n < 5) | (r == 0.0) & _degenerate(fa, fb, off, cc)
    p = np.where(short, 1.0, p)
    return r, p


def _degenerate(fa, fb, off, cols):
    out = np.zeros((len(off) - 1, len(cols)), bool)
    for s in range(len(off) - 1):
        a, b = off[s], off[s + 1]
        for k, c in enumerate(cols):
            out[s, k] = (b - a) < 5 or np.std(fa[a:b, c]) <= 1e-10 or np.std(fb[a:b, c]) <= 1e-10
    return out


def wasserstein_batch(rows_a, cnt_a, rows_b, cnt_b, idx_a=None, idx_b=None, ctx=None, want_status=False):
    ctx = ctx or get_ctx()
    ra = f64(rows_a); rb = f64(rows_b); ca = i32(cnt_a); cb = i32(cnt_b)
    n_a, cap_a, _ = ra.shape
    n_b, cap_b, _ = rb.shape
    if idx_a is None and idx_b is None:
        assert n_a == n_b
        n_pairs = n_a
    else:
        n_pairs = len(idx_a if idx_a is not None else idx_b)
    ia = None if idx_a is None else i32(idx_a)
    ib = None if idx_b is None else i32(idx_b)
    out = np.empty(n_pairs); st = np.empty(n_pairs, np.int32)
    ctx.check(ctx.lib.tda_wasserstein_batch(ctx.h, ptr(ra), ptr(ca), n_a, cap_a, ptr(rb), ptr(cb), n_b, cap_b,
                                            ptr(ia), ptr(ib), n_pairs, ptr(out), ptr(st)))
    return (out, st) if want_status else out


# ------------------------------------------------------------------ device-tensor API (torch)
def _tp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceDiagrams:
    """H0/H1 rows of a batch of windows, resident in HBM."""

    def __init__(self, n_win, h0_cap, h1_cap, device):
        import torch
        kw = dict(device=device)
        self.h0 = torch.empty((n_win, h0_cap, 2), dtype=torch.float64, **kw)
        self.h1 = torch.empty((n_win, h1_cap, 2), dtype=torch.float64, **kw)
        self.c0 = torch.empty(n_win, dtype=torch.int32, **kw)
        self.c1 = torch.empty(n_win, dtype=torch.int32, **kw)
        self.status = torch.empty(n_win, dtype=torch.int32, **kw)
        self.n_points = torch.empty(n_win, dtype=torch.int32, **kw)
        self.h0_cap, self.h1_cap, self.n_win = h0_cap, h1_cap, n_win

    def to_lists(self):
        h0, c0 = self.h0.cpu().numpy(), self.c0.cpu().numpy()
        h1, c1 = self.h1.cpu().numpy(), self.c1.cpu().numpy()
        return _diagrams(h0, c0, self.h0_cap), _diagrams(h1, c1, self.h1_cap)


def corr_dist_dev(win_t, dist_t=None, corr_t=None, ctx=None):
    import torch
    ctx = ctx or get_ctx()
    assert win_t.is_cuda and win_t.dtype == torch.float64 and win_t.is_contiguous()
    n_win, n_ch, n_t = win_t.shape
    if dist_t is None:
        dist_t = torch.empty((n_win, n_ch, n_ch), dtype=torch.float64, device=win_t.device)
    ctx.check(ctx.lib.tda_corr_dist_batch_dev(ctx.h, _tp(win_t), n_win, n_ch, n_t, _tp(dist_t), _tp(corr_t), _stream()))
    return dist_t


def rips_dm_dev(dm_t, out=None, thresh=MAX_EDGE_LENGTH, symmetrise=True, h1_cap=DEFAULT_H1_CAP, ctx=None):
    import torch
    ctx = ctx or get_ctx()
    assert dm_t.is_cuda and dm_t.dtype == torch.float64 and dm_t.is_contiguous()
    n_win, n, _ = dm_t.shape
    out = out or DeviceDiagrams(n_win, n, h1_cap, dm_t.device)
    ctx.check(ctx.lib.tda_rips_dm_batch_dev(ctx.h, _tp(dm_t), n_win, n, float(thresh), int(bool(symmetrise)),
                                            _tp(out.h0), out.h0_cap, _tp(out.c0), _tp(out.h1), out.h1_cap,
                                            _tp(out.c1), _tp(out.status), _stream()))
    return out


def eeg_window_dev(win_t, out=None, thresh=MAX_EDGE_LENGTH, h1_cap=DEFAULT_H1_CAP, dist_t=None, corr_t=None, ctx=None):
    """Fused corr -> dist -> Rips of EEG windows (nb2:198-207 + utils.py:135-141), one launch; the matrices are
    written only when dist_t (and corr_t) are given."""
    import torch
    ctx = ctx or get_ctx()
    assert win_t.is_cuda and win_t.dtype == torch.float64 and win_t.is_contiguous()
    n_win, n_ch, n_t = win_t.shape
    out = out or DeviceDiagrams(n_win, n_ch, h1_cap, win_t.device)
    ctx.check(ctx.lib.tda_eeg_window_batch_dev(ctx.h, _tp(win_t), n_win, n_ch, n_t, float(thresh), _tp(dist_t), _tp(corr_t),
                                               _tp(out.h0), out.h0_cap, _tp(out.c0), _tp(out.h1), out.h1_cap, _tp(out.c1),
                                               _tp(out.status), _stream()))
    return out


def eeg_window_sliding_dev(sig_t, win_len=250, step=62, sel_t=None, out=None, thresh=MAX_EDGE_LENGTH,
                           h1_cap=DEFAULT_H1_CAP, dist_t=None, corr_t=None, ctx=None):
    """Fused corr -> dist -> Rips on windows read in place from band-passed recordings sig_t (n_rec, n_ch,
    n_samples) (nb1:314-381 + nb2:198-207 + utils.py:135-141).  sel_t: optional int32 tensor of window indices
    r * n_win_per_rec + k.  Returns (diagrams, n_win_per_rec)."""
    import torch
    ctx = ctx or get_ctx()
    assert sig_t.is_cuda and sig_t.dtype == torch.float64 and sig_t.is_contiguous() and sig_t.dim() == 3
    n_rec, n_ch, n_s = sig_t.shape
    per_rec = (n_s - win_len) // step + 1 if n_s >= win_len else 0
    n_out = int(sel_t.numel()) if sel_t is not None else n_rec * per_rec
    out = out or DeviceDiagrams(n_out, n_ch, h1_cap, sig_t.device)
    assert out.n_win == n_out
    ctx.check(ctx.lib.tda_eeg_window_sliding_dev(ctx.h, _tp(sig_t), n_rec, n_ch, n_s, win_len, step, _tp(sel_t),
                                                 0 if sel_t is None else n_out, float(thresh), _tp(dist_t), _tp(corr_t),
                                                 _tp(out.h0), out.h0_cap, _tp(out.c0), _tp(out.h1), out.h1_cap, _tp(out.c1),
                                                 _tp(out.status), None, _stream()))
    return out, per_rec


def takens_rips_dev(win_t, tau_t, out=None, dim=3, subsample=2, thresh=MAX_EDGE_LENGTH, h1_cap=DEFAULT_H1_CAP,
                    ctx=None):
    import torch
    ctx = ctx or get_ctx()
    assert win_t.is_cuda and win_t.dtype == torch.float64 and win_t.is_contiguous()
    assert tau_t.dtype == torch.int32 and tau_t.is_cuda
    n_win, n_t = win_t.shape
    out = out or DeviceDiagrams(n_win, _lib.MAX_POINTS, h1_cap, win_t.device)
    ctx.check(ctx.lib.tda_takens_rips_batch_dev(ctx.h, _tp(win_t), _tp(tau_t), n_win, n_t, dim, subsample,
                                                float(thresh), _tp(out.h0), out.h0_cap, _tp(out.c0), _tp(out.h1),
                                                out.h1_cap, _tp(out.c1), _tp(out.n_points), _tp(out.status),
                                                _stream()))
    return out


def tau_dev(win_t, max_lag=None, tau_t=None, ctx=None):
    import torch
    ctx = ctx or get_ctx()
    n_win, n_t = win_t.shape
    if tau_t is None:
        tau_t = torch.empty(n_win, dtype=torch.int32, device=win_t.device)
    ctx.check(ctx.lib.tda_tau_batch_dev(ctx.h, _tp(win_t), n_win, n_t, -1 if max_lag is None else int(max_lag),
                                        _tp(tau_t), _stream()))
    return tau_t


def tau_segments_dev(win_t, seg_off_t, max_lag=None, tau_seg_t=None, tau_win_t=None, ctx=None):
    """cmp:83 / mvm:56: tau of every (recording, band) group from its first window; tau_win_t (optional, (n_win,)
    int32) receives the group's value for every window."""
    import torch
    ctx = ctx or get_ctx()
    n_win, n_t = win_t.shape
    n_seg = seg_off_t.numel() - 1
    if tau_seg_t is None:
        tau_seg_t = torch.empty(n_seg, dtype=torch.int32, device=win_t.device)
    ctx.check(ctx.lib.tda_tau_segments_dev(ctx.h, _tp(win_t), _tp(seg_off_t), n_seg, n_t,
                                           -1 if max_lag is None else int(max_lag), _tp(tau_seg_t), _tp(tau_win_t),
                                           _stream()))
    return tau_seg_t


def recording_rows_dev(w0_t, w1_t, tau_seg_t, fe0_t, fe1_t, seg_off_t, out_t=None, status_a=None, status_b=None,
                       seg_flags=None, ctx=None):
    """(n_seg, 48) rows [nanmean W_H0, nanmean W_H1, tau, n_windows, 44 aggregated features] in one launch.
    seg_flags (optional, (n_seg,) int32): per group, OR of the class-overflow bits of the two status arrays."""
    import torch
    ctx = ctx or get_ctx()
    n_seg = seg_off_t.numel() - 1
    if out_t is None:
        out_t = torch.empty((n_seg, 4 + 4 * _lib.N_FEATURES), dtype=torch.float64, device=w0_t.device)
    assert out_t.is_contiguous()
    ctx.check(ctx.lib.tda_recording_rows_dev(ctx.h, _tp(w0_t), _tp(w1_t), _tp(tau_seg_t), _tp(fe0_t), _tp(fe1_t),
                                             _tp(seg_off_t), n_seg, _tp(out_t), _tp(status_a), _tp(status_b),
                                             _tp(seg_flags), _stream()))
    return out_t


def features_dev(rows_t, cnt_t, feat_t=None, ctx=None):
    import torch
    ctx = ctx or get_ctx()
    n, cap, _ = rows_t.shape
    if feat_t is None:
        feat_t = torch.empty((n, _lib.N_FEATURES), dtype=torch.float64, device=rows_t.device)
    ctx.check(ctx.lib.tda_features_batch_dev(ctx.h, _tp(rows_t), _tp(cnt_t), n, cap, _tp(feat_t), _stream()))
    return feat_t


def diagram_finish_dev(sets, ctx=None):
    """ONE launch over up to four diagram sets: sets = [(rows_t (n, cap, 2), cnt_t (n,), order: bool, feat_t or
    None), ...] -- H1 rows into ripser's order where order is set, extract_features where feat_t is given."""
    ctx = ctx or get_ctx()
    arr = (_lib.DiagramSet * len(sets))()
    n = sets[0][0].shape[0]
    for i, (rows_t, cnt_t, order, feat_t) in enumerate(sets):
        assert rows_t.shape[0] == n and rows_t.is_contiguous()
        arr[i].rows = rows_t.data_ptr(); arr[i].cnt = cnt_t.data_ptr(); arr[i].cap = rows_t.shape[1]
        arr[i].order = int(bool(order)); arr[i].feat = feat_t.data_ptr() if feat_t is not None else None
    ctx.check(ctx.lib.tda_diagram_finish_dev(ctx.h, C.cast(arr, C.c_void_p), len(sets), n, _stream()))


def aggregate_dev(f0_t, f1_t, seg_off_t, out_t=None, ctx=None):
    import torch
    ctx = ctx or get_ctx()
    n_seg = seg_off_t.numel() - 1
    if out_t is None:
        out_t = torch.empty((n_seg, 4 * _lib.N_FEATURES), dtype=torch.float64, device=f0_t.device)
    ctx.check(ctx.lib.tda_aggregate_batch_dev(ctx.h, _tp(f0_t), _tp(f1_t), _tp(seg_off_t), n_seg, _tp(out_t),
                                              _stream()))
    return out_t


def segment_nanmean_dev(x_t, seg_off_t, out_t=None, ctx=None):
    import torch
    ctx = ctx or get_ctx()
    n_seg = seg_off_t.numel() - 1
    if out_t is None:
        out_t = torch.empty(n_seg, dtype=torch.float64, device=x_t.device)
    ctx.check(ctx.lib.tda_segment_nanmean_dev(ctx.h, _tp(x_t), _tp(seg_off_t), n_seg, _tp(out_t), _stream()))
    return out_t


def wasserstein_dev(rows_a, cnt_a, rows_b, cnt_b, idx_a=None, idx_b=None, out_t=None, status_t=None, ctx=None):
    import torch
    ctx = ctx or get_ctx()
    n_pairs = rows_a.shape[0] if idx_a is None and idx_b is None else (idx_a if idx_a is not None else idx_b).numel()
    if out_t is None:
        out_t = torch.empty(n_pairs, dtype=torch.float64, device=rows_a.device)
    if status_t is None:
        status_t = torch.empty(n_pairs, dtype=torch.int32, device=rows_a.device)
    ctx.check(ctx.lib.tda_wasserstein_batch_dev(ctx.h, _tp(rows_a), _tp(cnt_a), rows_a.shape[1], _tp(rows_b),
                                                _tp(cnt_b), rows_b.shape[1], _tp(idx_a), _tp(idx_b), n_pairs,
                                                _tp(out_t), _tp(status_t), _stream()))
    return out_t, status_t

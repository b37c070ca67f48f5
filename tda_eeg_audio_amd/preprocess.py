"""
preprocess.py -- front ends of the hot path ("next" rows 2-3), same names as
the reference:

  design_bandpass_filter, apply_bandpass_filter, create_sliding_windows
        notebooks/1_preprocesamiento.ipynb:209-263, 314-381  (EEG: zero-phase Butterworth per channel)
  bandpass_filter
        scripts/utils.py:66-74                               (audio envelope: filtfilt per band)
  eeg_to_distances
        preprocess_file (nb1:388-494) + process_file_graphs (nb2:158-218) fused: raw EEG (47, L) ->
        per band: band-pass on the GPU -> correlation/distance of the sliding windows read in place;
        neither the filtered windows stack (n_win, 47, 250) nor its 4x overlap ever exist.

Filter DESIGN (scipy.signal.butter, sosfilt_zi / lfilter_zi, pad length) is host-side preparation
done with scipy, exactly as the reference does; the recursions run in csrc/filters.hip and are
bit-identical to scipy.signal.sosfiltfilt / filtfilt.
"""
import ctypes as C

import numpy as np
from scipy import signal

from . import engine
from ._lib import f64, get_ctx, ptr

FREQ_BANDS = {"delta": (0.5, 4), "theta": (4, 8), "alpha": (8, 13), "beta": (13, 30), "gamma": (30, 50)}
FILTER_ORDER = 4            # nb1:128
WINDOW_SIZE_SEC = 1.0       # nb1:131
OVERLAP_PERCENT = 0.75      # nb1:132


def design_bandpass_filter(lowcut, highcut, fs, order=4):
    """nb1:209-233."""
    nyquist = 0.5 * fs
    return signal.butter(order, [lowcut / nyquist, highcut / nyquist], btype="band", output="sos")


def _sos_plan(sos):
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    n_sections = sos.shape[0]
    ntaps = 2 * n_sections + 1
    ntaps -= min((sos[:, 2] == 0).sum(), (sos[:, 5] == 0).sum())      # scipy.signal.sosfiltfilt
    return sos, np.ascontiguousarray(signal.sosfilt_zi(sos)), int(ntaps * 3)


def sosfiltfilt(sos, x, ctx=None):
    """scipy.signal.sosfiltfilt(sos, x, axis=-1) for a (n_sig, n_samples) float64 array."""
    ctx = ctx or get_ctx()
    x = f64(x)
    one = x.ndim == 1
    x2 = x.reshape(1, -1) if one else x
    sos, zi, edge = _sos_plan(sos)
    if x2.shape[1] <= edge:
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {edge}.")
    y = np.empty_like(x2)
    ctx.check(ctx.lib.tda_sosfiltfilt(ctx.h, ptr(x2), x2.shape[0], x2.shape[1], ptr(sos), ptr(zi), sos.shape[0], edge,
                                      ptr(y)))
    return y[0] if one else y


def filtfilt(b, a, x, ctx=None):
    """scipy.signal.filtfilt(b, a, x, axis=-1) (default odd padding) for (n_sig, n_samples) float64."""
    ctx = ctx or get_ctx()
    b = np.ascontiguousarray(np.atleast_1d(b), dtype=np.float64)
    a = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
    ntaps = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(ntaps - len(b))]); a = np.concatenate([a, np.zeros(ntaps - len(a))])
    zi = np.ascontiguousarray(signal.lfilter_zi(b, a))
    edge = 3 * ntaps
    x = f64(x)
    one = x.ndim == 1
    x2 = x.reshape(1, -1) if one else x
    if x2.shape[1] <= edge:
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {edge}.")
    y = np.empty_like(x2)
    ctx.check(ctx.lib.tda_filtfilt(ctx.h, ptr(x2), x2.shape[0], x2.shape[1], ptr(b), ptr(a), ptr(zi), ntaps, edge, ptr(y)))
    return y[0] if one else y


def apply_bandpass_filter(data, lowcut, highcut, fs, order=4):
    """nb1:236-263 -- all channels in one launch instead of a Python loop over channels."""
    return sosfiltfilt(design_bandpass_filter(lowcut, highcut, fs, order), np.asarray(data, dtype=np.float64))


def bandpass_filter(s, fs, low, high):
    """scripts/utils.py:66-74."""
    nyq = fs / 2
    lo = max(low / nyq, 0.001)
    hi = min(high / nyq, 0.999)
    if lo >= hi:
        return s
    b, a = signal.butter(4, [lo, hi], btype="band")
    return filtfilt(b, a, s)


def create_sliding_windows(data, window_size, overlap, fs):
    """nb1:314-381 -- (n_windows, n_channels, window_samples) stack and window centre times (host slicing;
    the GPU path never builds this stack, see eeg_to_distances)."""
    data = np.asarray(data)
    n_channels, n_samples = data.shape
    window_samples = int(window_size * fs)
    step_samples = int(window_samples * (1 - overlap))
    n_windows = (n_samples - window_samples) // step_samples + 1
    if n_windows <= 0:
        return np.zeros((0, n_channels, window_samples)), np.zeros(0)
    idx = np.arange(n_windows)[:, None] * step_samples + np.arange(window_samples)[None, :]
    windows = np.ascontiguousarray(data[:, idx].transpose(1, 0, 2))
    times = (np.arange(n_windows) * step_samples + window_samples // 2) / fs
    return windows, times


def bandpass_dev(x_t, lowcut, highcut, fs, order=FILTER_ORDER, y_t=None, work_t=None, ctx=None):
    """apply_bandpass_filter (nb1:236-263) on device tensors: x_t (n_sig, n_samples) float64 -- every channel of every
    recording of equal length in ONE launch; returns y_t (same shape).  work_t: optional (n_sig, n_samples + 2*edge)."""
    import torch
    ctx = ctx or get_ctx()
    assert x_t.is_cuda and x_t.dtype == torch.float64 and x_t.is_contiguous() and x_t.dim() == 2
    n_sig, n_s = x_t.shape
    sos, zi, edge = _sos_plan(design_bandpass_filter(lowcut, highcut, fs, order))
    if n_s <= edge:
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {edge}.")
    if y_t is None:
        y_t = torch.empty_like(x_t)
    if work_t is None or work_t.numel() < n_sig * (n_s + 2 * edge):
        work_t = torch.empty((n_sig, n_s + 2 * edge), dtype=torch.float64, device=x_t.device)
    ctx.check(ctx.lib.tda_sosfiltfilt_dev(ctx.h, C.c_void_p(x_t.data_ptr()), n_sig, n_s, ptr(sos), ptr(zi), sos.shape[0], edge,
                                          C.c_void_p(y_t.data_ptr()), C.c_void_p(work_t.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y_t


def recordings_to_features(raw_t, fs, sel_t=None, n_sel_per_rec=None, freq_bands=FREQ_BANDS, window_size=WINDOW_SIZE_SEC,
                           overlap=OVERLAP_PERCENT, order=FILTER_ORDER, ctx=None):
    """The EEG half of the corpus from RAW recordings, on the GPU end to end (preprocess_file nb1:388-494 +
    process_file_graphs nb2:158-218 + the hot loop and aggregation of process_file_features v2:404-436):
    raw_t (n_rec, n_ch, n_samples) float64 in HBM, recordings of equal length -> per band: zero-phase band-pass of
    all n_rec * n_ch channels in one launch -> fused window kernel on the sliding windows read in place (the
    (n_win, n_ch, 250) stacks and the distance matrices never exist) -> H1 row order + extract_features -> mean / std
    over each recording's windows.  sel_t: optional int32 window list (r * n_win_per_rec + k, n_sel_per_rec per
    recording, recording-major) -- the drivers' window selection.  Returns (n_rec, 44 * n_bands) float64 tensor in the
    column order of features/feature_names.txt, and the status words of the windows, OR-ed over the bands."""
    import torch
    from . import engine
    ctx = ctx or get_ctx()
    n_rec, n_ch, n_s = raw_t.shape
    win = int(window_size * fs)
    step = int(win * (1 - overlap))
    per_rec = (n_s - win) // step + 1 if n_s >= win else 0
    k = per_rec if sel_t is None else int(n_sel_per_rec)
    n_out = n_rec * k
    dev = raw_t.device
    flat = raw_t.view(n_rec * n_ch, n_s)
    y = torch.empty_like(flat)
    work = None
    dgm = engine.DeviceDiagrams(n_out, n_ch, engine.DEFAULT_H1_CAP, dev)
    fe0 = torch.empty((n_out, 11), dtype=torch.float64, device=dev)
    fe1 = torch.empty_like(fe0)
    seg = torch.arange(0, n_out + 1, k, dtype=torch.int32, device=dev)
    X = torch.empty((n_rec, len(freq_bands), 44), dtype=torch.float64, device=dev)
    status = torch.zeros(n_out, dtype=torch.int32, device=dev)
    ctx.set_h1_order(ctx.ORDER_DEFERRED)
    try:
        for bi, (name, (lo, hi)) in enumerate(freq_bands.items()):
            sos, zi, edge = _sos_plan(design_bandpass_filter(lo, hi, fs, order))
            if work is None:
                work = torch.empty((n_rec * n_ch, n_s + 2 * edge + 64), dtype=torch.float64, device=dev)
            bandpass_dev(flat, lo, hi, fs, order, y_t=y, work_t=work, ctx=ctx)
            engine.eeg_window_sliding_dev(y.view(n_rec, n_ch, n_s), win, step, sel_t=sel_t, out=dgm, ctx=ctx)
            engine.diagram_finish_dev([(dgm.h0, dgm.c0, False, fe0), (dgm.h1, dgm.c1, True, fe1)], ctx=ctx)
            X[:, bi].copy_(engine.aggregate_dev(fe0, fe1, seg, ctx=ctx))
            status |= dgm.status                 # (the next band overwrites the words)
    finally:
        ctx.set_h1_order(ctx.ORDER_IN_CALL)
    return X.view(n_rec, len(freq_bands) * 44), status


def filtfilt_dev(x_t, b, a, y_t=None, work_t=None, ctx=None):
    """scipy.signal.filtfilt(b, a, x, axis=-1) on device tensors: x_t (n_sig, n_samples) float64, one launch (the band-pass
    of the audio envelope, utils.py:66-74, for every recording of a shard at once)."""
    import torch
    ctx = ctx or get_ctx()
    assert x_t.is_cuda and x_t.dtype == torch.float64 and x_t.is_contiguous() and x_t.dim() == 2
    b = np.ascontiguousarray(np.atleast_1d(b), dtype=np.float64)
    a = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
    ntaps = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(ntaps - len(b))]); a = np.concatenate([a, np.zeros(ntaps - len(a))])
    zi = np.ascontiguousarray(signal.lfilter_zi(b, a))
    edge = 3 * ntaps
    n_sig, n_s = x_t.shape
    if n_s <= edge:
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {edge}.")
    if y_t is None:
        y_t = torch.empty_like(x_t)
    if work_t is None or work_t.numel() < n_sig * (n_s + 2 * edge):
        work_t = torch.empty((n_sig, n_s + 2 * edge), dtype=torch.float64, device=x_t.device)
    ctx.check(ctx.lib.tda_filtfilt_dev(ctx.h, C.c_void_p(x_t.data_ptr()), n_sig, n_s, ptr(b), ptr(a), ptr(zi), ntaps, edge,
                                       C.c_void_p(y_t.data_ptr()), C.c_void_p(work_t.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y_t


def bandpass_bank_dev(x_t, bands, fs, order=FILTER_ORDER, y_t=None, work_t=None, ctx=None):
    """apply_bandpass_filter (nb1:236-263) for ALL bands in one launch: x_t (n_sig, n_samples) -> y_t (n_bands, n_sig,
    n_samples), bit-identical to one bandpass_dev call per band.  bands: iterable of (lowcut, highcut)."""
    import torch
    ctx = ctx or get_ctx()
    assert x_t.is_cuda and x_t.dtype == torch.float64 and x_t.is_contiguous() and x_t.dim() == 2
    n_sig, n_s = x_t.shape
    plans = [_sos_plan(design_bandpass_filter(lo, hi, fs, order)) for lo, hi in bands]
    nf, n_sec, edge = len(plans), plans[0][0].shape[0], plans[0][2]
    assert all(p[0].shape[0] == n_sec and p[2] == edge for p in plans), "the filters of a bank share their structure"
    if n_s <= edge:
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {edge}.")
    sos = np.ascontiguousarray(np.stack([p[0] for p in plans])); zi = np.ascontiguousarray(np.stack([p[1] for p in plans]))
    if y_t is None:
        y_t = torch.empty((nf, n_sig, n_s), dtype=torch.float64, device=x_t.device)
    if work_t is None or work_t.numel() < nf * n_sig * (n_s + 2 * edge):
        work_t = torch.empty((nf, n_sig, n_s + 2 * edge), dtype=torch.float64, device=x_t.device)
    ctx.check(ctx.lib.tda_sosfiltfilt_bank_dev(ctx.h, C.c_void_p(x_t.data_ptr()), n_sig, n_s, ptr(sos), ptr(zi), nf, n_sec, edge,
                                               C.c_void_p(y_t.data_ptr()), C.c_void_p(work_t.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y_t


def filtfilt_bank_dev(x_t, bas, y_t=None, work_t=None, ctx=None):
    """scipy.signal.filtfilt for a bank of (b, a) pairs of equal length in one launch: x_t (n_sig, n_samples) -> y_t
    (n_filters, n_sig, n_samples) -- bandpass_filter (utils.py:66-74) of the audio envelope for all bands (cmp:63-64)."""
    import torch
    ctx = ctx or get_ctx()
    assert x_t.is_cuda and x_t.dtype == torch.float64 and x_t.is_contiguous() and x_t.dim() == 2
    n_sig, n_s = x_t.shape
    ntaps = max(max(len(b), len(a)) for b, a in bas)
    B = np.zeros((len(bas), ntaps)); A = np.zeros((len(bas), ntaps)); Z = np.zeros((len(bas), ntaps - 1))
    for f, (b, a) in enumerate(bas):
        B[f, :len(b)] = b; A[f, :len(a)] = a
        Z[f] = signal.lfilter_zi(B[f], A[f])
    edge = 3 * ntaps
    if n_s <= edge:
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {edge}.")
    if y_t is None:
        y_t = torch.empty((len(bas), n_sig, n_s), dtype=torch.float64, device=x_t.device)
    if work_t is None or work_t.numel() < len(bas) * n_sig * (n_s + 2 * edge):
        work_t = torch.empty((len(bas), n_sig, n_s + 2 * edge), dtype=torch.float64, device=x_t.device)
    ctx.check(ctx.lib.tda_filtfilt_bank_dev(ctx.h, C.c_void_p(x_t.data_ptr()), n_sig, n_s, ptr(B), ptr(A), ptr(Z), len(bas), ntaps,
                                            edge, C.c_void_p(y_t.data_ptr()), C.c_void_p(work_t.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y_t


def eeg_to_distances(eeg, fs, freq_bands=FREQ_BANDS, window_size=WINDOW_SIZE_SEC, overlap=OVERLAP_PERCENT,
                     order=FILTER_ORDER, want_corr=False, ctx=None):
    """raw EEG (n_ch, n_samples) -> {band: (n_win, n_ch, n_ch) distance matrices}; everything between the
    raw samples and the matrices stays in HBM (torch tensors, one stream)."""
    import torch
    ctx = ctx or get_ctx()
    dev = torch.device("cuda", ctx.device)
    x = torch.from_numpy(f64(eeg)).to(dev)
    n_ch, n_s = x.shape
    win = int(window_size * fs)
    step = int(win * (1 - overlap))
    out = {}
    for name, (lo, hi) in freq_bands.items():
        sos, zi, edge = _sos_plan(design_bandpass_filter(lo, hi, fs, order))
        y = torch.empty_like(x)
        work = torch.empty((n_ch, n_s + 2 * edge), dtype=torch.float64, device=dev)
        ctx.check(ctx.lib.tda_sosfiltfilt_dev(ctx.h, C.c_void_p(x.data_ptr()), n_ch, n_s, ptr(sos), ptr(zi), sos.shape[0],
                                              edge, C.c_void_p(y.data_ptr()), C.c_void_p(work.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        n_win = (n_s - win) // step + 1 if n_s >= win else 0
        if n_win <= 0:
            continue
        corr_t = torch.empty((n_win, n_ch, n_ch), dtype=torch.float64, device=dev) if want_corr else None
        dist_t = engine.corr_dist_sliding_dev(y, win, step, corr_t=corr_t, ctx=ctx)
        out[name] = (corr_t.cpu().numpy(), dist_t.cpu().numpy()) if want_corr else dist_t.cpu().numpy()
    return out


# --------------------------------------------------------------------------------------------
# audio front end (scripts/utils.py:47-79)
# --------------------------------------------------------------------------------------------
FS_AUDIO, FS_EEG = 44100, 250


def _resample_plan(n_in, up, down, window=("kaiser", 5.0)):
    """The FIR design and bookkeeping of scipy.signal.resample_poly (constant padding, cval 0)."""
    import math
    g_ = math.gcd(up, down)
    up //= g_; down //= g_
    n_out = n_in * up
    n_out = n_out // down + bool(n_out % down)
    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = signal.firwin(2 * half_len + 1, 1.0 / max_rate, window=window).astype(np.float64)
    h *= up
    n_pre_pad = down - half_len % down
    n_post_pad = 0
    n_pre_remove = (half_len + n_pre_pad) // down

    def _output_len(len_h, in_len, up_, down_):
        return (((in_len - 1) * up_ + len_h) - 1) // down_ + 1
    while _output_len(len(h) + n_pre_pad + n_post_pad, n_in, up, down) < n_out + n_pre_remove:
        n_post_pad += 1
    h = np.concatenate((np.zeros(n_pre_pad), h, np.zeros(n_post_pad)))
    return np.ascontiguousarray(h), up, down, n_pre_remove, n_out


def resample_audio(audio, fs_audio=FS_AUDIO, fs_target=FS_EEG, ctx=None):
    """scripts/utils.py:77-79 -- scipy.signal.resample_poly(audio, fs_target, fs_audio)."""
    ctx = ctx or get_ctx()
    x = f64(np.asarray(audio).ravel())
    if fs_audio == fs_target:
        return x.copy()
    h, up, down, n_pre_remove, n_out = _resample_plan(len(x), int(fs_target), int(fs_audio))
    y = np.empty(n_out)
    ctx.check(ctx.lib.tda_upfirdn(ctx.h, ptr(x), len(x), ptr(h), len(h), up, down, n_pre_remove, n_out, ptr(y)))
    return y


def hilbert_envelope(s, ctx=None):
    """np.abs(scipy.signal.hilbert(s)) (utils.py:58-59)."""
    ctx = ctx or get_ctx()
    x = f64(np.asarray(s).ravel())
    n = len(x)
    hh = np.zeros(n)
    if n % 2 == 0:
        hh[0] = hh[n // 2] = 1; hh[1:n // 2] = 2
    else:
        hh[0] = 1; hh[1:(n + 1) // 2] = 2
    g = np.ascontiguousarray(np.fft.ifft(hh).imag)
    env = np.empty(n)
    ctx.check(ctx.lib.tda_hilbert_envelope(ctx.h, ptr(x), n, ptr(g), ptr(env)))
    return env


def compute_envelope(s, fs, ctx=None):
    """scripts/utils.py:56-63 -- Hilbert envelope, then 4th-order low-pass (zero-phase)."""
    env = hilbert_envelope(s, ctx=ctx)
    nyq = fs / 2
    cutoff = min(50, nyq * 0.9)
    b, a = signal.butter(4, cutoff / nyq, btype="low")
    return filtfilt(b, a, env, ctx=ctx)


def audio_to_band_windows(audio, fs_audio=FS_AUDIO, freq_bands=FREQ_BANDS, ctx=None):
    """cmp:53-65 -- 44.1 kHz audio -> 250 Hz envelope -> per band zero-phase band-pass -> 1 s windows
    (step 62): {band: (n_win, 250)}."""
    from .utils import create_windows
    rs = resample_audio(audio, fs_audio, FS_EEG, ctx=ctx)
    env = compute_envelope(rs, FS_EEG, ctx=ctx)
    win = int(1.0 * FS_EEG)
    step = int(win * (1 - 0.75))
    return {name: create_windows(bandpass_filter(env, FS_EEG, lo, hi), win, step) for name, (lo, hi) in freq_bands.items()}

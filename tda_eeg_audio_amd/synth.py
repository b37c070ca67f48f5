"""
synth.py -- synthetic inputs of the shapes BASELINE.json names (no corpus is shipped:
data/, graphs/, preprocessed/ are git-ignored in the reference, .gitignore:9-16).

  eeg_windows   : (n_win, 47, 250) float64 with realistic correlation structure
                  X = A S + 0.5 E, 8 latent sources, A fixed per "recording" 
  audio_windows : (n_win, 250) float64 band-limited noise per frequency band so that the
                  autocorrelation delay tau lands in the observed ranges (config 4)
"""
import numpy as np

FREQ_BANDS = {"delta": (0.5, 4), "theta": (4, 8), "alpha": (8, 13), "beta": (13, 30), "gamma": (30, 50)}
N_CH, N_T, FS = 47, 250, 250


def eeg_windows(n_win, seed=42, n_ch=N_CH, n_t=N_T, kind="latent", windows_per_recording=71):
    out = np.empty((n_win, n_ch, n_t))
    A = None
    for w in range(n_win):
        rng = np.random.default_rng(seed + w)
        if kind == "white":
            out[w] = rng.standard_normal((n_ch, n_t))
            continue
        if w % windows_per_recording == 0 or A is None:
            A = np.random.default_rng(seed * 7919 + w // windows_per_recording).standard_normal((n_ch, 8))
        S = rng.standard_normal((8, n_t))
        out[w] = A @ S + 0.5 * rng.standard_normal((n_ch, n_t))
    return out


def audio_windows(n_win, band="gamma", seed=42, n_t=N_T):
    from scipy import signal
    lo, hi = FREQ_BANDS[band]
    nyq = FS / 2
    b, a = signal.butter(4, [max(lo / nyq, 0.001), min(hi / nyq, 0.999)], btype="band")
    step = 62
    rng = np.random.default_rng(seed)
    need = 500 + n_t + step * (n_win - 1) + 500
    x = signal.filtfilt(b, a, rng.standard_normal(need))
    return np.stack([x[500 + i * step: 500 + i * step + n_t] for i in range(n_win)]).copy()


def audio_windows_all_bands(n_per_band, seed=42):
    """(5*n_per_band, 250) windows, band-major, plus the band index of each window."""
    wins, band_id = [], []
    for bi, band in enumerate(FREQ_BANDS):
        wins.append(audio_windows(n_per_band, band, seed + 1000 * bi))
        band_id += [bi] * n_per_band
    return np.concatenate(wins), np.array(band_id)


# --------------------------------------------------------------------------------------------
# corpus-shaped workloads (BASELINE.json configs[2], configs[4])
# --------------------------------------------------------------------------------------------
BANDS = list(FREQ_BANDS)


def corpus_window_counts(n_rec=1416, seed=42):
    """Windows per recording with the corpus' shape (results/preprocessing_metadata.csv: 710 slow recordings
    with 65-89 windows, 706 fast ones with 39-54), slow ones first as scripts/tda_eeg_classification_v2.py:532-535
    lists them."""
    rng = np.random.default_rng(seed)
    n_slow = (n_rec * 710 + 1415) // 1416
    slow = rng.integers(65, 90, n_slow)
    fast = rng.integers(39, 55, n_rec - n_slow)
    return np.concatenate([slow, fast]).astype(np.int64)


def corpus_audio(n_rec, n_per_rec, bands=BANDS, seed=4242, n_t=N_T):
    """{band: (n_rec, n_per_rec, n_t) float64}: band-limited noise (Butterworth-4 at the band edges, zero phase), so
    that the delay tau of compute_tau lands in the observed ranges (gamma 2 ... delta 27-102)
    and the Takens clouds span 23 ... 123 points.  The SAME array on every rank (the recordings are dealt
    afterwards), so the total work does not depend on the number of GPUs."""
    from scipy import signal
    nyq = FS / 2
    out = {}
    stride = n_t - 2                                      # the selected windows of a recording hardly overlap
    for bi, band in enumerate(bands):
        lo, hi = FREQ_BANDS[band]
        b, a = signal.butter(4, [max(lo / nyq, 0.001), min(hi / nyq, 0.999)], btype="band")
        rng = np.random.default_rng(seed + 1000 * bi)
        need = 1000 + n_t + stride * (n_rec * n_per_rec - 1)
        x = signal.filtfilt(b, a, rng.standard_normal(need))
        idx = 500 + stride * np.arange(n_rec * n_per_rec)[:, None] + np.arange(n_t)[None, :]
        out[band] = np.ascontiguousarray(x[idx].reshape(n_rec, n_per_rec, n_t))
    return out


def corpus_eeg_dev(rec_ids, n_per_rec, n_bands, device, seed=42, n_ch=N_CH, n_t=N_T):
    """List (per band) of (len(rec_ids) * n_per_rec, n_ch, n_t) float64 tensors in HBM: X = A S + 0.5 E with 8
    latent sources and A fixed per recording (the `latent` kind of eeg_windows), drawn on the GPU from a
    generator seeded per RECORDING -- a recording's windows are the same whichever rank owns it.  torch is
    the random-number plumbing here; nothing of the hot path runs in it."""
    import torch
    f64 = dict(dtype=torch.float64, device=device)
    base = torch.empty((n_bands, len(rec_ids) * n_per_rec, n_ch, n_t), **f64)      # the bands lie back to back
    out = [base[b] for b in range(n_bands)]
    g = torch.Generator(device=device)
    for i, rec in enumerate(rec_ids):
        g.manual_seed(int(seed) * 1000003 + int(rec))
        A = torch.randn((n_ch, 8), generator=g, **f64)
        S = torch.randn((n_bands, n_per_rec, 8, n_t), generator=g, **f64)
        E = torch.randn((n_bands, n_per_rec, n_ch, n_t), generator=g, **f64)
        X = torch.matmul(A, S).add_(E, alpha=0.5)
        for b in range(n_bands):
            out[b][i * n_per_rec:(i + 1) * n_per_rec] = X[b]
    return out

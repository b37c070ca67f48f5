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

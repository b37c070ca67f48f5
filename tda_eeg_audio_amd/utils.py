"""
utils.py -- drop-in for the hot-path names of the reference's scripts/utils.py.

Same names, arguments and error behaviour (file:line of the reference in each docstring);
every computation runs in the HIP kernels of libtdaeeg.so through the C ABI.  A driver that
does ``from tda_eeg_audio_amd.utils import *`` instead of ``from utils import *`` keeps working.
Single-call functions launch a batch of one; for throughput use tda_eeg_audio_amd.engine /
pipeline, which keep whole batches resident in HBM.

Not provided here (host-side preparation in front of the path):
load_audio, compute_envelope, bandpass_filter, resample_audio (utils.py:47-79) and
permute_labels_by_subject (utils.py:198-215, statistics).
"""
import numpy as np

from . import engine
from ._lib import TdaError

# ── TDA parameters ── (scripts/utils.py:24-27)
MAX_DIM = 1
MAX_EDGE_LENGTH = 2.0
TAKENS_DIM = 3
TAKENS_SUBSAMPLE = 2

# ── Frequency bands ── (scripts/utils.py:30-36)
FREQ_BANDS = {
    "delta": (0.5, 4),
    "theta": (4, 8),
    "alpha": (8, 13),
    "beta": (13, 30),
    "gamma": (30, 50),
}

# ── Sampling rates ── (scripts/utils.py:39-40)
FS_AUDIO = 44100
FS_EEG = 250

FEATURE_KEYS = ["n_features", "n_essential", "mean_birth", "std_birth", "mean_death", "std_death",
                "mean_persistence", "std_persistence", "max_persistence", "total_persistence",
                "persistence_entropy"]


def create_windows(s, win_samples, step_samples):
    """scripts/utils.py:82-89 -- overlapping windows of a 1-D signal (pure slicing, host side)."""
    s = np.asarray(s)
    n = (len(s) - win_samples) // step_samples + 1 if len(s) >= win_samples else 0
    if n <= 0:
        return np.array([]).reshape(0, win_samples)
    idx = np.arange(n)[:, None] * step_samples + np.arange(win_samples)[None, :]
    return s[idx]


def compute_tau(s, max_lag=None):
    """scripts/utils.py:92-104 -- first zero crossing of the autocorrelation (tau_kernel)."""
    s = np.asarray(s, dtype=np.float64).reshape(1, -1)
    return int(engine.tau_batch(s, max_lag)[0])


def takens_embedding(s, dim, tau, subsample=1):
    """scripts/utils.py:107-116 -- the (P, dim) delay cloud.  A pure gather; the batched path
    (engine.takens_rips_batch) performs it inside the Rips kernel and never materialises it."""
    s = np.asarray(s)
    n = len(s) - (dim - 1) * tau
    if n <= 0:
        return np.array([]).reshape(0, dim)
    indices = np.arange(n)[:, None] + np.arange(dim)[None, :] * tau
    pc = s[indices]
    if subsample > 1:
        pc = pc[::subsample]
    return pc


def _check_dim(max_dim):
    if max_dim != 1:
        raise NotImplementedError("the HIP engine computes H0 and H1 (maxdim=1), as the reference does")


def compute_audio_persistence(point_cloud, max_dim=MAX_DIM, max_edge_length=MAX_EDGE_LENGTH):
    """scripts/utils.py:123-132 -- [H0, H1] of a point cloud (min-max normalised, Rips)."""
    _check_dim(max_dim)
    point_cloud = np.asarray(point_cloud, dtype=np.float64)
    if len(point_cloud) < 3:
        return [np.array([[0, 0]]), np.array([[0, 0]])]
    h0, h1, st = engine.cloud_rips_batch(point_cloud[None], thresh=max_edge_length, h1_cap=_h1_cap(len(point_cloud)))
    _check_status(st[0])
    return [h0[0], h1[0]]


def compute_eeg_persistence(dist_matrix, max_dim=MAX_DIM, max_edge_length=MAX_EDGE_LENGTH):
    """scripts/utils.py:135-141 -- [H0, H1] of a distance matrix (symmetrised, diag 0, >= 0)."""
    _check_dim(max_dim)
    dm = np.asarray(dist_matrix, dtype=np.float64)
    if dm.ndim != 2 or dm.shape[0] != dm.shape[1]:
        raise ValueError("Distance matrix is not square")      # the only thing ripser rejects
    h0, h1, st = engine.rips_dm_batch(dm[None], thresh=max_edge_length, symmetrise=True,
                                      h1_cap=_h1_cap(dm.shape[0]))
    _check_status(st[0])
    return [h0[0], h1[0]]


def _h1_cap(n):
    return max(256, n * (n - 1) // 2 - (n - 1)) if n <= 64 else 1024


def _check_status(st):
    if st & 2:
        raise TdaError("H1 class capacity exceeded; raise it with Context.set_class_words")
    if st & 1:
        raise TdaError("H1 diagram truncated; call engine.rips_dm_batch with a larger h1_cap")
    if st & 16:
        raise TdaError("more than 128 points: not supported by the LDS-resident kernels")


def extract_features(diagram):
    """scripts/utils.py:144-177 -- the 11 scalar features of one diagram (features_kernel)."""
    rows, cnt = engine.pack_diagrams([np.asarray(diagram, dtype=np.float64)])
    f = engine.features_batch(rows, cnt)[0]
    out = {k: float(v) for k, v in zip(FEATURE_KEYS, f)}
    out["n_features"] = int(f[0])
    out["n_essential"] = int(f[1])
    return out


def safe_wasserstein(dgm1, dgm2):
    """scripts/utils.py:180-191 -- persim.wasserstein on cleaned diagrams; NaN on any failure."""
    try:
        d1, d2 = np.asarray(dgm1), np.asarray(dgm2)
        ra, ca = engine.pack_diagrams([d1 if d1.ndim == 2 else np.zeros((0, 2))])
        rb, cb = engine.pack_diagrams([d2 if d2.ndim == 2 else np.zeros((0, 2))])
        out, st = engine.wasserstein_batch(ra, ca, rb, cb, want_status=True)
        return float(out[0]) if st[0] == 0 else np.nan
    except Exception:
        return np.nan

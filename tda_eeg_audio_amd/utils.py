"""
utils.py -- drop-in for the hot-path names of the reference's scripts/utils.py.

Same names, arguments and error behaviour (file:line of the reference in each docstring);
every computation runs in the HIP kernels of libtdaeeg.so through the C ABI.  A driver that
does ``from tda_eeg_audio_amd.utils import *`` instead of ``from utils import *`` keeps working.
Single-call functions launch a batch of one; ``with utils.batch():`` around the reference's per-window loop turns
the same calls into one launch per stage (see class batch); for throughput use tda_eeg_audio_amd.engine / pipeline,
which keep whole batches resident in HBM.

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
    if _ACTIVE is not None:
        return _ACTIVE.add_cloud(point_cloud, max_edge_length)
    h0, h1, st = engine.cloud_rips_batch(point_cloud[None], thresh=max_edge_length, h1_cap=_h1_cap(len(point_cloud)))
    _check_status(st[0])
    return [h0[0], h1[0]]


def compute_eeg_persistence(dist_matrix, max_dim=MAX_DIM, max_edge_length=MAX_EDGE_LENGTH):
    """scripts/utils.py:135-141 -- [H0, H1] of a distance matrix (symmetrised, diag 0, >= 0)."""
    _check_dim(max_dim)
    dm = np.asarray(dist_matrix, dtype=np.float64)
    if dm.ndim != 2 or dm.shape[0] != dm.shape[1]:
        raise ValueError("Distance matrix is not square")      # the only thing ripser rejects
    if _ACTIVE is not None:
        return _ACTIVE.add_dm(dm, max_edge_length)
    h0, h1, st = engine.rips_dm_batch(dm[None], thresh=max_edge_length, symmetrise=True,
                                      h1_cap=_h1_cap(dm.shape[0]))
    _check_status(st[0])
    return [h0[0], h1[0]]


def _h1_cap(n):
    return max(256, n * (n - 1) // 2 - (n - 1)) if n <= 64 else 1024


def _check_status(st):
    if st & 2:
        raise TdaError("H1 class overflow left after the full ladder (cannot happen for <= 128 points)")
    if st & 1:
        raise TdaError("H1 diagram truncated; call engine.rips_dm_batch with a larger h1_cap")
    if st & 16:
        raise TdaError("more than 128 points: not supported by the LDS-resident kernels")


def extract_features(diagram):
    """scripts/utils.py:144-177 -- the 11 scalar features of one diagram (features_kernel)."""
    if _ACTIVE is not None:
        return _ACTIVE.add_features(diagram)
    rows, cnt = engine.pack_diagrams([np.asarray(diagram, dtype=np.float64)])
    f = engine.features_batch(rows, cnt)[0]
    out = {k: float(v) for k, v in zip(FEATURE_KEYS, f)}
    out["n_features"] = int(f[0])
    out["n_essential"] = int(f[1])
    return out


def safe_wasserstein(dgm1, dgm2):
    """scripts/utils.py:180-191 -- persim.wasserstein on cleaned diagrams; NaN on any failure."""
    if _ACTIVE is not None:
        return _ACTIVE.add_pair(dgm1, dgm2)
    try:
        d1, d2 = np.asarray(dgm1), np.asarray(dgm2)
        ra, ca = engine.pack_diagrams([d1 if d1.ndim == 2 else np.zeros((0, 2))])
        rb, cb = engine.pack_diagrams([d2 if d2.ndim == 2 else np.zeros((0, 2))])
        out, st = engine.wasserstein_batch(ra, ca, rb, cb, want_status=True)
        return float(out[0]) if st[0] == 0 else np.nan
    except Exception:
        return np.nan


# --------------------------------------------------------------------------------------------
# batch(): the reference's per-window loop, unchanged, at one launch per stage
# --------------------------------------------------------------------------------------------
class _Deferred:
    """A result that exists once its batch has been flushed; every use of the value flushes."""
    __slots__ = ("_batch", "_value")

    def __init__(self, batch):
        self._batch, self._value = batch, None

    def _get(self):
        if self._value is None:
            self._batch.flush()
        return self._value


class DeferredArray(_Deferred):
    """One persistence diagram, (k, 2) float64, of a queued compute_*_persistence call."""
    __slots__ = ()

    def __array__(self, dtype=None, copy=None):
        v = self._get()
        return v if dtype is None else v.astype(dtype)

    def __len__(self):
        return len(self._get())

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())

    @property
    def shape(self):
        return self._get().shape

    @property
    def ndim(self):
        return 2

    def __repr__(self):
        return repr(self._get())


class DeferredScalar(_Deferred):
    """The value of a queued safe_wasserstein call."""
    __slots__ = ()

    def __float__(self):
        return float(self._get())

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._get(), dtype=dtype or np.float64)

    def __repr__(self):
        return repr(float(self))


class DeferredFeatures(_Deferred):
    """The dict of a queued extract_features call."""
    __slots__ = ()

    def __getitem__(self, k):
        return self._get()[k]

    def keys(self):
        return self._get().keys()

    def items(self):
        return self._get().items()

    def values(self):
        return self._get().values()

    def __iter__(self):
        return iter(self._get())

    def __len__(self):
        return len(self._get())

    def __repr__(self):
        return repr(self._get())


_ACTIVE = None


class batch:
    """``with utils.batch():`` around the reference's per-window loop (scripts/tda_eeg_audio_comparison.py:88-99,
    scripts/matched_vs_mismatched.py:57-63,87-95) -- the loop stays as it is; compute_audio_persistence,
    compute_eeg_persistence, safe_wasserstein and extract_features queue their arguments and hand back deferred results,
    and on leaving the block (or at the first use of a value) everything queued runs as ONE launch per stage: the point
    clouds of all windows, the distance matrices of all windows, all Wasserstein pairs, all feature vectors.  Values,
    error behaviour (NaN from safe_wasserstein, [[0, 0]] for degenerate clouds, ValueError for a non-square matrix) and
    result types after the block are those of the immediate calls."""

    def __init__(self):
        self.clouds, self.dms, self.pairs, self.feats = [], [], [], []

    def __enter__(self):
        global _ACTIVE
        if _ACTIVE is not None:
            raise RuntimeError("utils.batch() blocks do not nest")
        _ACTIVE = self
        return self

    def __exit__(self, et, ev, tb):
        global _ACTIVE
        _ACTIVE = None
        if et is None:
            self.flush()
        return False

    # ---- queueing (called by the module functions while the block is active)
    def add_cloud(self, pc, thresh):
        d = [DeferredArray(self), DeferredArray(self)]
        self.clouds.append((pc, float(thresh), d))
        return d

    def add_dm(self, dm, thresh):
        d = [DeferredArray(self), DeferredArray(self)]
        self.dms.append((dm, float(thresh), d))
        return d

    def add_pair(self, a, b):
        s = DeferredScalar(self)
        self.pairs.append((a, b, s))
        return s

    def add_features(self, dgm):
        f = DeferredFeatures(self)
        self.feats.append((dgm, f))
        return f

    @staticmethod
    def _resolve(x):
        return x._value if isinstance(x, _Deferred) else x

    def flush(self):
        clouds, dms, pairs, feats = self.clouds, self.dms, self.pairs, self.feats
        self.clouds, self.dms, self.pairs, self.feats = [], [], [], []
        # ---- stage 1: all Rips calls (one launch per kernel flavour, threshold and matrix size)
        for th in sorted({c[1] for c in clouds}):
            grp = [c for c in clouds if c[1] == th]
            p_cap, dim = max(len(c[0]) for c in grp), grp[0][0].shape[1]
            pcs = np.zeros((len(grp), p_cap, dim))
            for i, c in enumerate(grp):
                pcs[i, :len(c[0])] = c[0]
            h0, h1, st = engine.cloud_rips_batch(pcs, n_pts=[len(c[0]) for c in grp], thresh=th, h1_cap=_h1_cap(p_cap))
            for i, c in enumerate(grp):
                _check_status(st[i])
                c[2][0]._value, c[2][1]._value = h0[i], h1[i]
        for key in sorted({(d[0].shape[0], d[1]) for d in dms}):
            grp = [d for d in dms if (d[0].shape[0], d[1]) == key]
            h0, h1, st = engine.rips_dm_batch(np.stack([d[0] for d in grp]), thresh=key[1], symmetrise=True, h1_cap=_h1_cap(key[0]))
            for i, d in enumerate(grp):
                _check_status(st[i])
                d[2][0]._value, d[2][1]._value = h0[i], h1[i]
        # ---- stage 2: all Wasserstein pairs, all feature vectors
        if pairs:
            def clean(x):
                x = np.asarray(self._resolve(x))
                return x if x.ndim == 2 else np.zeros((0, 2))
            try:
                ra, ca = engine.pack_diagrams([clean(p[0]) for p in pairs])
                rb, cb = engine.pack_diagrams([clean(p[1]) for p in pairs])
                out, st = engine.wasserstein_batch(ra, ca, rb, cb, want_status=True)
                for i, p in enumerate(pairs):
                    p[2]._value = float(out[i]) if st[i] == 0 else np.nan
            except Exception:                       # utils.py:190-191: any failure -> NaN
                for p in pairs:
                    p[2]._value = np.nan
        if feats:
            rows, cnt = engine.pack_diagrams([np.asarray(self._resolve(f[0]), dtype=np.float64) for f in feats])
            F = engine.features_batch(rows, cnt)
            for i, f in enumerate(feats):
                out = {k: float(v) for k, v in zip(FEATURE_KEYS, F[i])}
                out["n_features"] = int(F[i][0])
                out["n_essential"] = int(F[i][1])
                f[1]._value = out

"""
oracle/port.py -- ctypes binding of oracle/tda_oracle.c (libtda_oracle.so).

TEST INFRASTRUCTURE ONLY (see the header of tda_oracle.c).  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtda_oracle.so")
_SO_NATIVE = os.path.join(_HERE, "libtda_oracle_native.so")
_lib = None

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int)


def build(force=False):
    src = os.path.join(_HERE, "tda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_corr_dist.argtypes = [c_dp, C.c_int, C.c_int, c_dp, c_dp]
        L.orc_corr_dist.restype = None
        L.orc_corr_dist_batch.argtypes = [c_dp, C.c_int, C.c_int, C.c_int, c_dp, c_dp]
        L.orc_corr_dist_batch.restype = None
        L.orc_count_windows.argtypes = [C.c_int] * 3
        L.orc_compute_tau.argtypes = [c_dp, C.c_int, C.c_int]
        L.orc_takens.argtypes = [c_dp, C.c_int, C.c_int, C.c_int, C.c_int, c_dp]
        L.orc_minmax_normalise.argtypes = [c_dp, C.c_int, C.c_int, c_dp]
        L.orc_minmax_normalise.restype = None
        L.orc_cloud_dm.argtypes = [c_dp, C.c_int, C.c_int, c_dp]
        L.orc_cloud_dm.restype = None
        L.orc_rips_f32.argtypes = [c_fp, C.c_int, C.c_float, c_fp, C.c_int, c_ip, c_fp, C.c_int, c_ip]
        L.orc_rips_dm.argtypes = [c_dp, C.c_int, C.c_double, C.c_int, c_fp, C.c_int, c_ip, c_fp, C.c_int, c_ip]
        L.orc_rips_dm_batch.argtypes = [c_dp, C.c_int, C.c_int, C.c_double, C.c_int,
                                        c_fp, C.c_int, c_ip, c_fp, C.c_int, c_ip]
        L.orc_audio_persistence.argtypes = [c_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                            c_fp, C.c_int, c_ip, c_fp, C.c_int, c_ip, c_ip]
        L.orc_features.argtypes = [c_dp, C.c_int, c_dp]
        L.orc_features.restype = None
        L.orc_wasserstein.argtypes = [c_dp, C.c_int, c_dp, C.c_int]
        L.orc_wasserstein.restype = C.c_double
        L.orc_eeg_prepare.argtypes = [c_dp, C.c_int, C.c_int, c_fp]
        L.orc_eeg_prepare.restype = None
        _bind_segment(L)
        _lib = L
    return _lib


def _bind_segment(L):
    L.orc_segment_step.argtypes = [c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, c_dp]
    L.orc_segment_step.restype = C.c_int


def use_native():
    """Timing leg only (bench.py cpu_baseline): rebuild the SAME source with -O3 -march=native on this
    host and bind it; falls back to the portable build when the compiler is missing.  Returns the flags."""
    global _lib
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
        L = C.CDLL(_SO_NATIVE)
        _bind_segment(L)
        lib()                       # the portable one stays bound for everything else
        _lib.orc_segment_step = L.orc_segment_step
        _lib._native = L
        return "gcc -O3 -march=native"
    except Exception:
        lib()
        return "gcc -O3"


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(t)


def corr_dist(window):
    w = _d(window)
    n, t = w.shape
    corr = np.empty((n, n)); dist = np.empty((n, n))
    lib().orc_corr_dist(_p(w, c_dp), n, t, _p(corr, c_dp), _p(dist, c_dp))
    return corr, dist


def corr_dist_batch(windows):
    w = _d(windows)
    b, n, t = w.shape
    corr = np.empty((b, n, n)); dist = np.empty((b, n, n))
    lib().orc_corr_dist_batch(_p(w, c_dp), b, n, t, _p(corr, c_dp), _p(dist, c_dp))
    return corr, dist


def compute_tau(s, max_lag=None):
    s = _d(s)
    return int(lib().orc_compute_tau(_p(s, c_dp), len(s), -1 if max_lag is None else int(max_lag)))


def takens(s, dim, tau, subsample=1):
    s = _d(s)
    n = len(s) - (dim - 1) * tau
    cap = max(n, 1)
    out = np.empty((cap, dim))
    P = lib().orc_takens(_p(s, c_dp), len(s), dim, tau, subsample, _p(out, c_dp))
    return out[:P].copy()


def minmax_normalise(pc):
    pc = _d(pc)
    out = np.empty_like(pc)
    lib().orc_minmax_normalise(_p(pc, c_dp), pc.shape[0], pc.shape[1], _p(out, c_dp))
    return out


def cloud_dm(pc_norm):
    x = _d(pc_norm)
    P, dim = x.shape
    dm = np.empty((P, P))
    lib().orc_cloud_dm(_p(x, c_dp), P, dim, _p(dm, c_dp))
    return dm


def _dgms(h0, k0, h1, k1):
    return [h0[:k0].astype(np.float64), h1[:k1].astype(np.float64)]


def rips_f32(dm_f32, thresh=2.0, h1_cap=None):
    f = np.ascontiguousarray(dm_f32, dtype=np.float32)
    n = f.shape[0]
    h1_cap = h1_cap or max(n * (n - 1) // 2, 1)
    h0 = np.empty((max(n, 1), 2), np.float32); h1 = np.empty((h1_cap, 2), np.float32)
    k0 = C.c_int(); k1 = C.c_int()
    st = lib().orc_rips_f32(_p(f, c_fp), n, thresh, _p(h0, c_fp), max(n, 1), C.byref(k0),
                            _p(h1, c_fp), h1_cap, C.byref(k1))
    assert st == 0
    return _dgms(h0, k0.value, h1, k1.value)


def rips_dm(dist, thresh=2.0, symmetrise=True):
    """compute_eeg_persistence (scripts/utils.py:135-141) on the oracle."""
    d = _d(dist)
    n = d.shape[0]
    cap1 = max(n * (n - 1) // 2, 1)
    h0 = np.empty((max(n, 1), 2), np.float32); h1 = np.empty((cap1, 2), np.float32)
    k0 = C.c_int(); k1 = C.c_int()
    st = lib().orc_rips_dm(_p(d, c_dp), n, thresh, int(symmetrise), _p(h0, c_fp), max(n, 1), C.byref(k0),
                           _p(h1, c_fp), cap1, C.byref(k1))
    assert st == 0
    return _dgms(h0, k0.value, h1, k1.value)


def rips_dm_batch(dist, thresh=2.0, symmetrise=True, h1_cap=256):
    d = _d(dist)
    b, n, _ = d.shape
    h0 = np.empty((b, n, 2), np.float32); h1 = np.empty((b, h1_cap, 2), np.float32)
    k0 = np.empty(b, np.int32); k1 = np.empty(b, np.int32)
    st = lib().orc_rips_dm_batch(_p(d, c_dp), b, n, thresh, int(symmetrise), _p(h0, c_fp), n, _p(k0, c_ip),
                                 _p(h1, c_fp), h1_cap, _p(k1, c_ip))
    return st, h0, k0, h1, k1


def audio_persistence(window, tau, dim=3, subsample=2, thresh=2.0):
    """takens_embedding + compute_audio_persistence (scripts/utils.py:107-132)."""
    s = _d(window)
    cap = 130 * 129 // 2
    h0 = np.empty((260, 2), np.float32); h1 = np.empty((cap, 2), np.float32)
    k0 = C.c_int(); k1 = C.c_int(); npts = C.c_int()
    st = lib().orc_audio_persistence(_p(s, c_dp), len(s), dim, int(tau), subsample, thresh,
                                     _p(h0, c_fp), 260, C.byref(k0), _p(h1, c_fp), cap, C.byref(k1),
                                     C.byref(npts))
    assert st == 0
    return _dgms(h0, k0.value, h1, k1.value), npts.value


FEATURE_KEYS = ["n_features", "n_essential", "mean_birth", "std_birth", "mean_death", "std_death",
                "mean_persistence", "std_persistence", "max_persistence", "total_persistence",
                "persistence_entropy"]


def features(dgm):
    d = _d(np.asarray(dgm).reshape(-1, 2))
    out = np.empty(11)
    lib().orc_features(_p(d, c_dp), d.shape[0], _p(out, c_dp))
    return out


def wasserstein(a, b):
    a = _d(np.asarray(a).reshape(-1, 2)); b = _d(np.asarray(b).reshape(-1, 2))
    return float(lib().orc_wasserstein(_p(a, c_dp), a.shape[0], _p(b, c_dp), b.shape[0]))


def segment_step(eeg_win, audio_win, max_lag=125, thresh=2.0):
    """One (recording, band) group end to end in C (orc_segment_step): the 48-value result row."""
    e = _d(eeg_win); a = _d(audio_win)
    n_win, n_ch, n_t = e.shape
    row = np.empty(48)
    st = lib().orc_segment_step(_p(e, c_dp), _p(a, c_dp), n_win, n_ch, n_t, int(max_lag), float(thresh), _p(row, c_dp))
    assert st == 0
    return row

"""
_lib.py -- ctypes binding of libtdaeeg.so (the C ABI in include/tdaeeg.h).

The HIP library is the product; there is NO CPU fallback.  If the shared object is
missing or no MI355X is visible, every entry point raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C tda_eeg_audio_amd/csrc``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TDA_LIB=libtdaeeg_<variant>.so: a diagnostic / A-B build from csrc/Makefile (same directory); the product library otherwise
LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("TDA_LIB", "libtdaeeg.so")))

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_vp = C.c_void_p

TDA_WIN_H1_TRUNCATED = 1
TDA_WIN_CLASS_OVERFLOW = 2
TDA_WIN_DEGENERATE = 4
TDA_WIN_NOT_CONVERGED = 8
TDA_WIN_TOO_LARGE = 16
N_FEATURES = 11
MAX_POINTS = 128

# every symbol include/tdaeeg.h declares: (name, restype, argtypes)
_I, _D = C.c_int, C.c_double
SYMBOLS = {
    "tda_version": (_I, []),
    "tda_ctx_create": (_I, [_I, C.POINTER(c_vp)]),
    "tda_ctx_destroy": (None, [c_vp]),
    "tda_last_error": (C.c_size_t, [c_vp, C.c_char_p, C.c_size_t]),
    "tda_set_class_words": (_I, [c_vp, _I, _I]),
    "tda_corr_dist_batch_dev": (_I, [c_vp, c_vp, _I, _I, _I, c_vp, c_vp, c_vp]),
    "tda_corr_dist_batch": (_I, [c_vp, c_vp, _I, _I, _I, c_vp, c_vp]),
    "tda_corr_dist_sliding_dev": (_I, [c_vp, c_vp, _I, _I, _I, _I, c_vp, c_vp, c_vp, c_vp]),
    "tda_corr_dist_sliding": (_I, [c_vp, c_vp, _I, _I, _I, _I, c_vp, c_vp, c_vp]),
    "tda_corr_to_dist_batch_dev": (_I, [c_vp, c_vp, _I, _I, _I, c_vp, c_vp]),
    "tda_corr_to_dist_batch": (_I, [c_vp, c_vp, _I, _I, _I, c_vp]),
    "tda_rips_dm_batch_dev": (_I, [c_vp, c_vp, _I, _I, _D, _I, c_vp, _I, c_vp, c_vp, _I, c_vp, c_vp, c_vp]),
    "tda_eeg_window_batch_dev": (_I, [c_vp, c_vp, _I, _I, _I, _D, c_vp, c_vp, c_vp, _I, c_vp, c_vp, _I, c_vp, c_vp, c_vp]),
    "tda_eeg_window_sliding_dev": (_I, [c_vp, c_vp, _I, _I, _I, _I, _I, c_vp, _I, _D, c_vp, c_vp, c_vp, _I, c_vp, c_vp, _I,
                                        c_vp, c_vp, c_vp, c_vp]),
    "tda_rips_dm_batch": (_I, [c_vp, c_vp, _I, _I, _D, _I, c_vp, _I, c_vp, c_vp, _I, c_vp, c_vp]),
    "tda_takens_rips_batch_dev": (_I, [c_vp, c_vp, c_vp, _I, _I, _I, _I, _D, c_vp, _I, c_vp, c_vp, _I, c_vp,
                                       c_vp, c_vp, c_vp]),
    "tda_takens_rips_batch": (_I, [c_vp, c_vp, c_vp, _I, _I, _I, _I, _D, c_vp, _I, c_vp, c_vp, _I, c_vp,
                                   c_vp, c_vp]),
    "tda_cloud_rips_batch_dev": (_I, [c_vp, c_vp, c_vp, _I, _I, _I, _I, _D, c_vp, _I, c_vp, c_vp, _I, c_vp,
                                      c_vp, c_vp]),
    "tda_cloud_rips_batch": (_I, [c_vp, c_vp, c_vp, _I, _I, _I, _I, _D, c_vp, _I, c_vp, c_vp, _I, c_vp, c_vp]),
    "tda_sosfiltfilt_dev": (_I, [c_vp, c_vp, _I, _I, c_vp, c_vp, _I, _I, c_vp, c_vp, c_vp]),
    "tda_sosfiltfilt": (_I, [c_vp, c_vp, _I, _I, c_vp, c_vp, _I, _I, c_vp]),
    "tda_filtfilt_dev": (_I, [c_vp, c_vp, _I, _I, c_vp, c_vp, c_vp, _I, _I, c_vp, c_vp, c_vp]),
    "tda_filtfilt": (_I, [c_vp, c_vp, _I, _I, c_vp, c_vp, c_vp, _I, _I, c_vp]),
    "tda_sosfiltfilt_bank_dev": (_I, [c_vp, c_vp, _I, _I, c_vp, c_vp, _I, _I, _I, c_vp, c_vp, c_vp]),
    "tda_filtfilt_bank_dev": (_I, [c_vp, c_vp, _I, _I, c_vp, c_vp, c_vp, _I, _I, _I, c_vp, c_vp, c_vp]),
    "tda_upfirdn_dev": (_I, [c_vp, c_vp, C.c_longlong, c_vp, _I, _I, _I, C.c_longlong, C.c_longlong, c_vp, c_vp]),
    "tda_upfirdn": (_I, [c_vp, c_vp, C.c_longlong, c_vp, _I, _I, _I, C.c_longlong, C.c_longlong, c_vp]),
    "tda_hilbert_envelope_dev": (_I, [c_vp, c_vp, _I, c_vp, c_vp, c_vp]),
    "tda_hilbert_envelope": (_I, [c_vp, c_vp, _I, c_vp, c_vp]),
    "tda_tau_batch_dev": (_I, [c_vp, c_vp, _I, _I, _I, c_vp, c_vp]),
    "tda_tau_segments_dev": (_I, [c_vp, c_vp, c_vp, _I, _I, _I, c_vp, c_vp, c_vp]),
    "tda_recording_rows_dev": (_I, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, _I, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tda_set_retry_policy": (_I, [c_vp, _I]),
    "tda_set_retry_counter": (_I, [c_vp, c_vp]),
    "tda_set_h1_order": (_I, [c_vp, _I]),
    "tda_diagram_finish_dev": (_I, [c_vp, c_vp, _I, _I, c_vp]),
    "tda_tau_batch": (_I, [c_vp, c_vp, _I, _I, _I, c_vp]),
    "tda_features_batch_dev": (_I, [c_vp, c_vp, c_vp, _I, _I, c_vp, c_vp]),
    "tda_features_batch": (_I, [c_vp, c_vp, c_vp, _I, _I, c_vp]),
    "tda_aggregate_batch_dev": (_I, [c_vp, c_vp, c_vp, c_vp, _I, c_vp, c_vp]),
    "tda_aggregate_batch": (_I, [c_vp, c_vp, c_vp, c_vp, _I, _I, c_vp]),
    "tda_segment_nanmean_dev": (_I, [c_vp, c_vp, c_vp, _I, c_vp, c_vp]),
    "tda_segment_nanmean": (_I, [c_vp, c_vp, c_vp, _I, _I, c_vp]),
    "tda_spearman_batch_dev": (_I, [c_vp, c_vp, c_vp, _I, c_vp, _I, c_vp, _I, c_vp, c_vp]),
    "tda_spearman_batch": (_I, [c_vp, c_vp, c_vp, _I, _I, c_vp, _I, c_vp, _I, c_vp]),
    "tda_wasserstein_batch_dev": (_I, [c_vp, c_vp, c_vp, _I, c_vp, c_vp, _I, c_vp, c_vp, _I, c_vp, c_vp, c_vp]),
    "tda_wasserstein_batch": (_I, [c_vp, c_vp, c_vp, _I, _I, c_vp, c_vp, _I, _I, c_vp, c_vp, _I, c_vp, c_vp]),
    "tda_event_create": (_I, [c_vp, C.POINTER(c_vp)]),
    "tda_event_record": (_I, [c_vp, c_vp, c_vp]),
    "tda_event_elapsed_ms": (_I, [c_vp, c_vp, c_vp, C.POINTER(C.c_float)]),
    "tda_event_destroy": (_I, [c_vp, c_vp]),
    "tda_set_kernel_probe": (_I, [c_vp, _I, c_vp, c_vp, c_vp]),
    "tda_stream_sync": (_I, [c_vp, c_vp]),
}

class DiagramSet(C.Structure):
    """tda_diagram_set of include/tdaeeg.h."""
    _fields_ = [("rows", c_vp), ("cnt", c_vp), ("cap", _I), ("order", _I), ("feat", c_vp)]


_lib = None


class TdaError(RuntimeError):
    pass


def load():
    """dlopen libtdaeeg.so.  torch (if importable) is imported first so that both share ONE HIP
    runtime (both link libamdhip64.so.7; the first one loaded wins)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TdaError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch's bundled HIP runtime first)
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Context:
    """One per process and GPU (tda_ctx)."""

    def __init__(self, device=0):
        self.lib = load()
        h = c_vp()
        rc = self.lib.tda_ctx_create(int(device), C.byref(h))
        if rc != 0:
            buf = C.create_string_buffer(512)
            self.lib.tda_last_error(None, buf, 512)
            raise TdaError(f"tda_ctx_create(device={device}) failed ({rc}): {buf.value.decode()} "
                           "-- the HIP path needs a visible MI355X; there is no CPU fallback")
        self.h = h
        self.device = int(device)
        # first-pass class capacity (x64 bits) of the Rips kernels, e.g. TDA_CLASS_WORDS=1,1 (dm, cloud)
        cw = os.environ.get("TDA_CLASS_WORDS")
        if cw:
            dm, cloud = (int(x) for x in cw.split(","))
            self.set_class_words(dm, cloud)

    def check(self, rc):
        if rc != 0:
            buf = C.create_string_buffer(1024)
            self.lib.tda_last_error(self.h, buf, 1024)
            raise TdaError(f"libtdaeeg error {rc}: {buf.value.decode()}")

    def set_class_words(self, words_dm=2, words_cloud=1):
        self.check(self.lib.tda_set_class_words(self.h, words_dm, words_cloud))

    RETRY_AUTO, RETRY_FIRST_PASS, RETRY_ONLY, RETRY_ONE_STEP, RETRY_LAST_RUNG = 0, 1, 2, 3, 4

    def set_retry_policy(self, policy):
        self.check(self.lib.tda_set_retry_policy(self.h, int(policy)))

    ORDER_IN_CALL, ORDER_DEFERRED = 0, 1

    def set_h1_order(self, policy):
        self.check(self.lib.tda_set_h1_order(self.h, int(policy)))

    def set_retry_counter(self, dev_ptr):
        """dev_ptr: device address of a zeroed u64[4] (or None): windows redone by the widening passes."""
        self.check(self.lib.tda_set_retry_counter(self.h, c_vp(dev_ptr) if dev_ptr else None))

    # ---- one-shot kernel probe (bench.py roofline): HIP events around ONE first-pass kernel ----
    PROBES = {"rips_audio": 1, "rips_eeg": 2, "corr_dist": 3}

    def new_event(self):
        ev = c_vp()
        self.check(self.lib.tda_event_create(self.h, C.byref(ev)))
        return ev

    def arm_probe(self, stage, ev_start=None, ev_stop=None, dev_span=None):
        """dev_span: optional device pointer (int) to a zeroed u64[4] (rips_audio only): the kernel
        accumulates its own duration there (see include/tdaeeg.h); events may be omitted then."""
        self.check(self.lib.tda_set_kernel_probe(self.h, self.PROBES[stage], ev_start, ev_stop,
                                                 c_vp(dev_span) if dev_span else None))

    def elapsed_ms(self, ev_start, ev_stop):
        ms = C.c_float()
        self.check(self.lib.tda_event_elapsed_ms(self.h, ev_start, ev_stop, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tda_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_ctx = {}


def get_ctx(device=None):
    if device is None:
        device = int(os.environ.get("TDA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        try:
            import torch
            if torch.cuda.is_available():
                device = torch.cuda.current_device()
        except Exception:
            pass
    if device not in _ctx:
        _ctx[device] = Context(device)
    return _ctx[device]


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_vp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)

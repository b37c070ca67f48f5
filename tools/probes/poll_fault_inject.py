"""
Guard build only (libtdaeeg_dbg.so): the bounded LDS polls of the Rips sweep turn a broken dependency into a
status bit.  tda_debug_inject(1): wave 3 never hands on the turn of phase a; tda_debug_inject(2): one apparent edge
per chunk waits for itself.  Every audio window must come back with TDA_WIN_NOT_CONVERGED (8) -- not hang -- and,
with the fault switched off again, with status 0 and the right diagrams.
    python tools/probes/poll_fault_inject.py            (on the GPU box; tests/test_gpu_stress.py runs it)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib            # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg_dbg.so")
from tda_eeg_audio_amd import engine, synth   # noqa: E402

ctx = _lib.get_ctx(0)
ctx.lib.tda_debug_inject.argtypes = [_lib.C.c_int]
aw = synth.audio_windows(24, "beta", seed=11)
tau = int(engine.tau_batch(aw[:1], 125, ctx=ctx)[0])
ref = engine.takens_rips_batch(aw, tau, ctx=ctx)
assert not np.any(ref[3] & ~4), ref[3]
bad = 0
for what in (1, 2):
    assert ctx.lib.tda_debug_inject(what) == 0
    t0 = time.time()
    h0, h1, npts, st = engine.takens_rips_batch(aw, tau, ctx=ctx)
    dt = time.time() - t0
    # fault 2 sits on one lane: a window in which that lane never holds a dependent edge is untouched -- and must be right
    hit = (st & 8) != 0
    clean = all(np.array_equal(np.sort(h1[w], axis=0), np.sort(ref[1][w], axis=0)) for w in range(len(st)) if not hit[w])
    ok = bool(hit.all()) if what == 1 else bool(hit.any() and clean and not np.any(st & ~(4 | 8)))
    print(f"fault {what}: {int(hit.sum())} of {len(st)} windows report status 8 in {dt:.2f} s ->", "reported" if ok else "NOT reported", flush=True)
    bad += not ok
assert ctx.lib.tda_debug_inject(0) == 0
h0, h1, npts, st = engine.takens_rips_batch(aw, tau, ctx=ctx)
same = all(np.array_equal(np.sort(a, axis=0), np.sort(b, axis=0)) for a, b in zip(h1, ref[1])) and not np.any(st & ~4)
print("fault off again:", "clean" if same else "DIFFERENT")
bad += not same
print("POLL GUARD", "OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)

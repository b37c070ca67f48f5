"""Randomised parity stress of the Rips / Wasserstein kernels against the CPU oracle: tests/stress_cases.py at any
scale and seed (TDA_STRESS_SCALE, TDA_STRESS_SEED).  TDA_STRESS_DEBUG=1 loads the guard build
(make -C tda_eeg_audio_amd/csrc DEBUG_PTS=1): sentinel words between the LDS regions of the Rips kernels, checked
after the sweep (status bit 0x100), and the normalised cloud copied out of LDS before and after it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tda_eeg_audio_amd import _lib
DEBUG = os.environ.get("TDA_STRESS_DEBUG") == "1"
if DEBUG:
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "libtdaeeg_dbg.so")
from tda_eeg_audio_amd import engine, synth
import stress_cases

ctx = _lib.get_ctx(0)
s = stress_cases.Stress(ctx, engine, synth, seed=int(os.environ.get("TDA_STRESS_SEED", "2024")),
                        scale=int(os.environ.get("TDA_STRESS_SCALE", "1")), guard=DEBUG,
                        log=lambda *a: print(*a, flush=True))
what = os.environ.get("TDA_STRESS_WHAT", "rips,wasserstein").split(",")
if "rips" in what:
    s.rips_all()
if "clouds" in what:                      # the part the guard build is for, alone (quick)
    for w in ((2, 1), (1, 1)):
        ctx.set_class_words(*w); s.clouds(); s.ties(reps=2)
    ctx.set_class_words(2, 1)
if "wasserstein" in what:
    s.wasserstein()
    print("wasserstein done, mismatches so far:", s.bad, "(bar 1e-7)", flush=True)
print("STRESS", "OK" if s.bad == 0 else f"FAILED ({s.bad})")
sys.exit(1 if s.bad else 0)

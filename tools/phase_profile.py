#!/usr/bin/env python3
"""Phase-cycle breakdown of the Rips kernels (diagnostic build: make -C tda_eeg_audio_amd/csrc PROFILE=1)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg_prof.so")
from tda_eeg_audio_amd import engine, synth
ctx = _lib.get_ctx(0)
lib = ctx.lib
buf = (C.c_ulonglong * 16)()
for band in ["beta", "delta"]:
    wins = synth.audio_windows(256, band, seed=1)
    tau = engine.tau_batch(wins[:1], 125, ctx=ctx)[0]
    engine.takens_rips_batch(wins, tau, ctx=ctx)
    lib.tda_profile_read(buf, 1)
    engine.takens_rips_batch(wins, tau, ctx=ctx)
    lib.tda_profile_read(buf, 1)
    v = np.array(list(buf), dtype=np.float64)
    n = v[8]
    print(f"audio {band} tau={tau}: windows={int(n)} E/win={v[9]/n:.0f} cycles/win: keygen={v[0]/n:.0f} sort={v[1]/n:.0f} "
          f"compact={v[2]/n:.0f} sweep={v[3]/n:.0f} (of which kill-path={v[4]/n:.0f}; kills/win={v[10]/n:.1f}, kill episodes/win={v[11]/n:.1f})")

#!/usr/bin/env python3
"""Instruction counts of the fused EEG window kernel by phase (diagnostic build; see tools/inst_split.py)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg_prof.so")
from tda_eeg_audio_amd import engine, synth
ctx = _lib.get_ctx(0); lib = ctx.lib
NW = 2048; dev = torch.device("cuda", 0)
ctx.set_retry_policy(ctx.RETRY_FIRST_PASS); ctx.set_h1_order(ctx.ORDER_DEFERRED); ctx.set_class_words(1, 1)
wins = torch.from_numpy(synth.eeg_windows(NW, seed=1, windows_per_recording=15)).to(dev)
out = engine.DeviceDiagrams(NW, 47, 256, dev)
for stop in (1, 2, 11, 12, 13, 14, 15, 16, 17, 18, 19, 0):
    lib.tda_profile_stop_after(stop)
    engine.eeg_window_dev(wins, out, ctx=ctx); torch.cuda.synchronize()
lib.tda_profile_stop_after(0)

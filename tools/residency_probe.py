#!/usr/bin/env python3
"""What would a third audio workgroup per CU buy?  Clouds of 98 points (delta band, tau = 27) fit a 49 KB layout when
the library is built with -DTDA_EXPERIMENT -DCLOUD_NB=4096 -DCLOUD_WAVES=6 and run with TDA_EXP_PMAX=98; the standard
layout (sized for 124 points) holds two workgroups per CU.  Usage: residency_probe.py <lib.so> [n_windows]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", sys.argv[1])
from tda_eeg_audio_amd import engine, synth
ctx = _lib.get_ctx(0)
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 40960
dev = torch.device("cuda", 0)
ctx.set_class_words(1, 1); ctx.set_retry_policy(ctx.RETRY_FIRST_PASS); ctx.set_h1_order(ctx.ORDER_DEFERRED)
wins = torch.from_numpy(synth.audio_windows(NW, "delta", seed=1)).to(dev)
tau_t = torch.full((NW,), 27, dtype=torch.int32, device=dev)
out = engine.DeviceDiagrams(NW, 128, 256, dev)
for _ in range(2):
    engine.takens_rips_dev(wins, tau_t, out, ctx=ctx)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    engine.takens_rips_dev(wins, tau_t, out, ctx=ctx)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print(f"{sys.argv[1]} PMAX={os.environ.get('TDA_EXP_PMAX')}: {NW} windows of {int(out.n_points[0])} points in {ms:.3f} ms = {NW / ms / 1e3:.3f} M windows/s; "
      f"flagged {int((out.status != 0).sum())}; h1 rows {int(out.c1.sum())}")

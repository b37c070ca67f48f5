#!/usr/bin/env python3
"""Instruction counts of the audio Rips kernel by phase: the diagnostic build returns after the keys / after the
ranking / runs all; run under  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace  and
read the three dispatches of rips_cloud_kernel per band in order."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg_prof.so")
from tda_eeg_audio_amd import engine, synth
ctx = _lib.get_ctx(0); lib = ctx.lib
NW = 2048; dev = torch.device("cuda", 0)
ctx.set_retry_policy(ctx.RETRY_FIRST_PASS); ctx.set_h1_order(ctx.ORDER_DEFERRED)
for band in ["beta", "delta"]:
    wins = torch.from_numpy(synth.audio_windows(NW, band, seed=1)).to(dev)
    tau = engine.tau_batch(wins[:1].cpu().numpy(), 125, ctx=ctx)[0]
    tau_t = torch.full((NW,), int(tau), dtype=torch.int32, device=dev)
    out = engine.DeviceDiagrams(NW, 128, 256, dev)
    stops = [1, 2, 11, 12, 13, 14, 15, 16, 17, 18, 19, 0]
    if os.environ.get("TDA_SPLIT_CHUNKS"):       # cumulative counts at the end of chunk 0..n-1, then everything
        stops = [1, 2] + [20 + 100 * c for c in range(int(os.environ["TDA_SPLIT_CHUNKS"]))] + [0]
    if os.environ.get("TDA_SPLIT_PHASES_OF"):    # the phases of chunk k (k >= 1), cumulative from the end of chunk k - 1
        k = int(os.environ["TDA_SPLIT_PHASES_OF"])
        stops = [20 + 100 * (k - 1)] + [n + 100 * k for n in range(11, 21)]
    for stop in stops:
        lib.tda_profile_stop_after(stop)
        engine.takens_rips_dev(wins, tau_t, out, ctx=ctx); torch.cuda.synchronize()
lib.tda_profile_stop_after(0)

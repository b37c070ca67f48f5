#!/usr/bin/env python3
"""Effective shader clock under the Rips kernels: cycles per window (diagnostic build, s_memtime) against the wall
time of the same launch (HIP events).  usage: clock_check.py [n_windows]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
if os.environ.get("TDA_PROF", "1") == "1":
    _lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg_prof.so")
from tda_eeg_audio_amd import engine, synth
ctx = _lib.get_ctx(0); lib = ctx.lib
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 21240
dev = torch.device("cuda", 0)
buf = (C.c_ulonglong * 32)()
for band in ["beta", "delta"]:
    wins = torch.from_numpy(synth.audio_windows(NW, band, seed=1)).to(dev)
    tau = engine.tau_batch(wins[:1].cpu().numpy(), 125, ctx=ctx)[0]
    tau_t = torch.full((NW,), int(tau), dtype=torch.int32, device=dev)
    out = engine.DeviceDiagrams(NW, 128, 256, dev)
    ctx.set_retry_policy(ctx.RETRY_FIRST_PASS)
    for rep in range(3):
        if hasattr(lib, "tda_profile_read"): lib.tda_profile_read(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.takens_rips_dev(wins, tau_t, out, ctx=ctx); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        msg = f"audio {band} tau={tau}: {NW} windows in {ms:.3f} ms = {NW/ms/1e3:.3f} M windows/s"
        if hasattr(lib, "tda_profile_read"):
            lib.tda_profile_read(buf, 1)
            v = np.array(list(buf), dtype=np.float64); n = v[8]
            cyc = (v[0] + v[1] + v[2] + v[3]) / n
            msg += f"; {cyc:.0f} cycles/window -> {cyc * NW / 512 / (ms * 1e-3) / 1e9:.2f} GHz if 512 windows were resident all the time"
        print(msg)
    ctx.set_retry_policy(ctx.RETRY_AUTO)

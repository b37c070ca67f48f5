#!/usr/bin/env python3
"""Launch-duration of the two Rips kernels alone vs number of windows (workgroup residency probe)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib, engine, synth
dev = torch.device("cuda", 0); ctx = _lib.get_ctx(0)
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
N = 2048
aud = torch.from_numpy(synth.audio_windows(N, "beta", seed=2)).to(dev)
tau = torch.full((N,), 3, dtype=torch.int32, device=dev)
W = torch.from_numpy(synth.eeg_windows(N, seed=1)).to(dev)
dist = engine.corr_dist_dev(W, ctx=ctx)
for n in (128, 256, 384, 512, 768, 1024, 2048):
    out = engine.DeviceDiagrams(n, 128, 256, dev)
    ta = timeit(lambda: engine.takens_rips_dev(aud[:n].contiguous(), tau[:n].contiguous(), out, ctx=ctx))
    oute = engine.DeviceDiagrams(n, 47, 256, dev)
    te = timeit(lambda: engine.rips_dm_dev(dist[:n].contiguous(), oute, ctx=ctx))
    print(f"n={n:5d}  audio {ta:7.3f} ms ({ta/n*256*1e3:6.1f} us per window-slot@256CU)   eeg {te:7.3f} ms")

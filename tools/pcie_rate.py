#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench step: host (pinned) windows -> HBM -> run_step -> result rows back.
Reported in DESIGN.md only; bench.py's `value` is measured with inputs resident in HBM."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib, pipeline, synth
dev = torch.device("cuda", 0); ctx = _lib.get_ctx(0)
for n_win in (710, 11360):
    seg = np.array(list(range(0, n_win, 15)) + [n_win], np.int32)
    eeg = torch.from_numpy(synth.eeg_windows(n_win, seed=1, windows_per_recording=15)).pin_memory()
    aud = torch.from_numpy(synth.audio_windows(n_win, "beta", seed=2)).pin_memory()
    eeg_d = torch.empty_like(eeg, device=dev); aud_d = torch.empty_like(aud, device=dev)
    ws = pipeline.Workspace(n_win, seg, dev)
    out_h = torch.empty((len(seg) - 1, pipeline.RESULT_COLS), dtype=torch.float64).pin_memory()
    def step():
        eeg_d.copy_(eeg, non_blocking=True); aud_d.copy_(aud, non_blocking=True)
        res = pipeline.run_step(eeg_d, aud_d, ws, ctx=ctx)
        out_h.copy_(res, non_blocking=True)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 10
    for _ in range(K): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"{n_win} windows: {dt*1e3:.3f} ms/step incl. H2D of {eeg.numel()*8/1e6:.1f} MB -> {n_win/dt:,.0f} windows/s "
          f"({eeg.numel()*8/dt/1e9:.1f} GB/s over PCIe)")

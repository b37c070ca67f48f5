"""Wall-time cost of each stage group inside the overlapped step: the bench loop with stage groups stubbed out
(outputs then hold stale data; timing only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tda_eeg_audio_amd import _lib, engine, pipeline, synth

dev = torch.device("cuda", 0)
ctx = _lib.get_ctx(0)
ctx.set_class_words(1, 1)
n_win, wpr = 710, 15
seg = np.array(list(range(0, n_win, wpr)) + [n_win], np.int32)
eeg_t = torch.from_numpy(synth.eeg_windows(n_win, seed=42, windows_per_recording=wpr)).to(dev)
aud_t = torch.from_numpy(synth.audio_windows(n_win, "beta", seed=4242)).to(dev)
REAL = {k: getattr(engine, k) for k in dir(engine) if k.endswith("_dev")}


def run(label, stub, lanes=3, steps=100):
    for k, v in REAL.items():
        setattr(engine, k, v)
    for k in stub:
        real = REAL[k]
        if k == "aggregate_dev":
            setattr(engine, k, lambda *a, **kw: torch.zeros((len(seg) - 1, 44), dtype=torch.float64, device=dev))
        elif k == "segment_nanmean_dev":
            setattr(engine, k, lambda *a, **kw: torch.zeros(len(seg) - 1, dtype=torch.float64, device=dev))
        else:
            setattr(engine, k, lambda *a, **kw: None)
    L = pipeline.Lanes(lanes, n_win, seg, dev, graph=True)
    for _ in range(2 * lanes):
        L.submit(eeg_t, aud_t, ctx=ctx, sync_inputs=False)          # real data in every buffer first
    L.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        L.submit(eeg_t, aud_t, ctx=ctx, sync_inputs=False)
    L.drain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{label:58s} {dt * 1e3:.4f} ms/step")


run("full step", [])
run("without EEG chain (corr, rips_dm, features_eeg, aggregate)", ["corr_dist_dev", "rips_dm_dev"])
run("without Wasserstein", ["wasserstein_dev"])
run("without features", ["features_dev"])
run("audio Rips only", ["corr_dist_dev", "rips_dm_dev", "wasserstein_dev", "features_dev", "aggregate_dev", "segment_nanmean_dev", "tau_segments_dev", "recording_rows_dev"])
run("without audio Rips", ["takens_rips_dev"])

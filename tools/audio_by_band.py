"""The audio Rips stage band by band (21,240 windows each): which clouds cost what.  python tools/audio_by_band.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib, engine, synth
ctx = _lib.get_ctx(0); dev = torch.device("cuda", 0)
ctx.set_class_words(1, 1); ctx.set_retry_policy(ctx.RETRY_ONE_STEP); ctx.set_h1_order(ctx.ORDER_DEFERRED)
aud = synth.corpus_audio(1416, 15)
for b in synth.BANDS:
    W = torch.from_numpy(aud[b].reshape(-1, 250)).to(dev)
    n = W.shape[0]
    seg = torch.arange(0, n + 1, 15, dtype=torch.int32, device=dev)
    tau_seg = torch.empty(n // 15, dtype=torch.int32, device=dev); tau_win = torch.empty(n, dtype=torch.int32, device=dev)
    engine.tau_segments_dev(W, seg, 125, tau_seg, tau_win, ctx=ctx)
    out = engine.DeviceDiagrams(n, 128, engine.DEFAULT_H1_CAP, dev)
    for _ in range(2):
        engine.takens_rips_dev(W, tau_win, out, ctx=ctx)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); engine.takens_rips_dev(W, tau_win, out, ctx=ctx); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e))
    npts = out.n_points.cpu().numpy()
    print(f"{b:6s} tau {int(tau_seg.min())}..{int(tau_seg.max())}  points {npts.min()}..{npts.max()} (mean {npts.mean():.0f}): {min(ts):.3f} ms per {n} windows "
          f"= {min(ts) / n * 1e3:.3f} us per window; h1 rows per window {float(out.c1.float().mean()):.1f}")

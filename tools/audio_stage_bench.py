"""The audio Rips stage alone on the bench's mix of clouds (all five bands, tau per recording-band as bench.py has it):
first pass + one widening pass, HIP events.   python tools/audio_stage_bench.py [lib.so] [n_rec]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", sys.argv[1])
from tda_eeg_audio_amd import engine, synth
n_rec = int(sys.argv[2]) if len(sys.argv) > 2 else 472
ctx = _lib.get_ctx(0); dev = torch.device("cuda", 0)
ctx.set_class_words(1, 1)
ctx.set_retry_policy(ctx.RETRY_ONE_STEP)
ctx.set_h1_order(ctx.ORDER_DEFERRED)
aud = synth.corpus_audio(n_rec, 15)
W = torch.from_numpy(np.concatenate([aud[b].reshape(-1, 250) for b in synth.BANDS])).to(dev)
n = W.shape[0]
seg = torch.arange(0, n + 1, 15, dtype=torch.int32, device=dev)
tau_seg = torch.empty(n // 15, dtype=torch.int32, device=dev); tau_win = torch.empty(n, dtype=torch.int32, device=dev)
engine.tau_segments_dev(W, seg, 125, tau_seg, tau_win, ctx=ctx)
out = engine.DeviceDiagrams(n, 128, engine.DEFAULT_H1_CAP, dev)
for _ in range(2):
    engine.takens_rips_dev(W, tau_win, out, ctx=ctx)
torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get('TDA_STAGE_REPS', '6'))):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); engine.takens_rips_dev(W, tau_win, out, ctx=ctx); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
bad = int(((out.status & ~4) != 0).sum())
print("launches (ms):", " ".join(f"{t:.2f}" for t in ts))
print(f"{os.path.basename(_lib.LIB_PATH)}: {n} audio windows (5 bands): {min(ts):.3f} ms = {n / min(ts) / 1e3:.3f} M windows/s "
      f"[{min(ts) * 106200 / n:.2f} ms per 106,200]; status left non-zero: {bad}; h1 rows {int(out.c1.sum())}")

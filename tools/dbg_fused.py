import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import brute, port
from tda_eeg_audio_amd import _lib, engine, synth
ctx = _lib.get_ctx(0); dev = torch.device("cuda", 0)
same = lambda a, b: np.array_equal(brute.sort_rows(a), brute.sort_rows(b))
W = synth.eeg_windows(96, seed=5, kind="white"); wt = torch.from_numpy(W).to(dev)
dist = engine.corr_dist_dev(wt, ctx=ctx)
two = engine.rips_dm_dev(dist, ctx=ctx)
ctx.set_class_words(1, 1)
ctx.set_retry_policy(ctx.RETRY_FIRST_PASS); first = engine.eeg_window_dev(wt, ctx=ctx); torch.cuda.synchronize()
fl = first.status.cpu().numpy()
ctx.set_retry_policy(ctx.RETRY_AUTO)
for rep in range(3):
    one = engine.eeg_window_dev(wt, ctx=ctx); torch.cuda.synchronize()
    a0, a1 = one.to_lists(); b0, b1 = two.to_lists(); st = one.status.cpu().numpy()
    d = dist.cpu().numpy()
    for w in range(96):
        o = port.rips_dm(d[w])
        f_ok = same(a0[w], o[0]) and same(a1[w], o[1]); t_ok = same(b0[w], o[0]) and same(b1[w], o[1])
        if not (f_ok and t_ok):
            print("rep", rep, "window", w, "first-pass flag", fl[w], "status", st[w], "fused ok", f_ok, "two-kernel ok", t_ok, len(a1[w]), len(b1[w]), len(o[1]))
print("flagged in first pass:", int((fl & 2).astype(bool).sum()))

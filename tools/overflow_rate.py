"""How many windows of the bench data run out of class bits in the FIRST pass, per first-pass capacity?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tda_eeg_audio_amd import _lib, engine, synth
ctx = _lib.get_ctx(0)
dev = torch.device("cuda", 0)
n = 2840
eeg = torch.from_numpy(synth.eeg_windows(n, seed=42, windows_per_recording=15)).to(dev)
dist = engine.corr_dist_dev(eeg, ctx=ctx)
ctx.set_retry_policy(ctx.RETRY_FIRST_PASS)
for words in (1, 2):
    ctx.set_class_words(words, 1)
    out = engine.rips_dm_dev(dist, ctx=ctx)
    torch.cuda.synchronize()
    st = out.status.cpu().numpy()
    print(f"EEG (latent-source windows): {64*words} class bits: {(st & 2 != 0).sum()} of {n} windows overflow the first pass")
for band in ("delta", "theta", "alpha", "beta", "gamma"):
    aud = torch.from_numpy(synth.audio_windows(1420, band, seed=4242)).to(dev)
    tau = engine.tau_dev(aud[:1].contiguous(), 125, ctx=ctx)
    tw = tau.repeat(1420).contiguous()
    out = engine.DeviceDiagrams(1420, 128, engine.DEFAULT_H1_CAP, dev)
    ctx.set_class_words(2, 1)
    engine.takens_rips_dev(aud, tw, out, ctx=ctx)
    torch.cuda.synchronize()
    st = out.status.cpu().numpy()
    print(f"audio {band} (tau={int(tau[0])}): 32 class bits: {(st & 2 != 0).sum()} of 1420 windows overflow the first pass")
ctx.set_retry_policy(ctx.RETRY_AUTO); ctx.set_class_words(2, 1)

"""How many audio workgroups does a CU hold?  Times the first pass of the point-cloud Rips kernel on N = k * 256 copies
of ONE window (identical work per workgroup): the time steps up whenever N passes a multiple of the resident capacity
(256 CUs x workgroups per CU).  python tools/residency_steps.py [lib.so]   (TDA_CLOUD_WIDE_FIRST=1: the wide layout)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", sys.argv[1])
from tda_eeg_audio_amd import engine, synth
ctx = _lib.get_ctx(0); dev = torch.device("cuda", 0)
ctx.set_class_words(1, 1)
ctx.set_retry_policy(ctx.RETRY_FIRST_PASS)
w = synth.audio_windows(1, "beta", seed=3)
for k in (1, 2, 3, 4, 5, 6, 8, 9, 12):
    n = 256 * k
    W = torch.from_numpy(np.repeat(w, n, axis=0)).to(dev)
    tau = torch.full((n,), 3, dtype=torch.int32, device=dev)
    out = engine.takens_rips_dev(W, tau, ctx=ctx)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.takens_rips_dev(W, tau, out, ctx=ctx); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"N = {n:5d} ({k:2d} per CU): {min(ts) * 1e3:8.1f} us   status {int(out.status.max())}", flush=True)

"""
Repro + discriminator for the fault of the fused EEG kernel's widening passes at __launch_bounds__(256, 1)
(rips.hip, eeg_window_kernel): is it the covariance (MFMA accumulators in AGPRs) or the sweep that goes wrong?

    make -C tda_eeg_audio_amd/csrc VARIANT=w1 EXTRA=-DTDA_EEG_WIDE_WAVES=1
    python tools/probes/wide_waves_repro.py libtdaeeg_w1.so        (on the GPU box)

Every window is flagged by hand and redone by the widening kernels only (TDA_RETRY_ONLY), with the optional
matrix outputs switched on: dist/corr are compared with corr_dist_kernel's (bit for bit), the diagrams with the
two-kernel path.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib, engine, synth      # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", sys.argv[1])
ctx = _lib.get_ctx(0)
dev = torch.device("cuda", 0)
N = 32
W = synth.eeg_windows(N, seed=5, kind="white")
wt = torch.from_numpy(W).to(dev)
dist_ref = torch.empty((N, 47, 47), dtype=torch.float64, device=dev)
corr_ref = torch.empty_like(dist_ref)
engine.corr_dist_dev(wt, dist_ref, corr_ref, ctx=ctx)
two = engine.rips_dm_dev(dist_ref, ctx=ctx)
torch.cuda.synchronize()
r0, r1 = two.to_lists()


def srt(a):
    a = np.asarray(a).reshape(-1, 2)
    return a[np.lexsort((a[:, 1], a[:, 0]))]


for words_dm, label in ((1, "128-bit widening pass (W=2, RETRY)"), (2, "512-bit widening pass (W=8, RETRY)")):
    ctx.set_class_words(words_dm, 1)
    out = engine.DeviceDiagrams(N, 47, engine.DEFAULT_H1_CAP, dev)
    out.status.fill_(2)                                  # every window "ran out of class bits"
    out.c0.zero_(); out.c1.zero_()
    dist = torch.full((N, 47, 47), -7.0, dtype=torch.float64, device=dev)
    corr = torch.full((N, 47, 47), -7.0, dtype=torch.float64, device=dev)
    ctx.set_retry_policy(ctx.RETRY_ONLY)
    engine.eeg_window_dev(wt, out, dist_t=dist, corr_t=corr, ctx=ctx)
    ctx.set_retry_policy(ctx.RETRY_AUTO)
    torch.cuda.synchronize()
    dd = (dist - dist_ref).abs().amax(dim=(1, 2)).cpu().numpy()
    dc = (corr - corr_ref).abs().amax(dim=(1, 2)).cpu().numpy()
    a0, a1 = out.to_lists()
    st = out.status.cpu().numpy()
    bad0 = [w for w in range(N) if not np.array_equal(srt(a0[w]), srt(r0[w]))]
    bad1 = [w for w in range(N) if not np.array_equal(srt(a1[w]), srt(r1[w]))]
    print(f"{label}: status!=0 {int((st != 0).sum())}/{N}; dist wrong in {int((dd > 0).sum())} windows (max |d| {dd.max():.3e}), "
          f"corr wrong in {int((dc > 0).sum())} (max {dc.max():.3e}); H0 wrong {len(bad0)}, H1 wrong {len(bad1)}")
    if bad0 or bad1:
        w = (bad0 + bad1)[0]
        print("  first bad window", w, "status", st[w], "rows H0", len(a0[w]), "vs", len(r0[w]), " H1", len(a1[w]), "vs", len(r1[w]))
        x, y = srt(a0[w]), srt(r0[w])
        n = min(len(x), len(y))
        k = np.nonzero((x[:n] != y[:n]).any(axis=1))[0]
        print("  H0 first differing rows:", x[k[:3]].tolist(), "ref", y[k[:3]].tolist())
        if (dd[w] > 0):
            d = (dist[w] - dist_ref[w]).abs().cpu().numpy()
            i, j = np.unravel_index(d.argmax(), d.shape)
            print("  dist worst entry", (int(i), int(j)), float(dist[w, i, j]), "ref", float(dist_ref[w, i, j]),
                  " rows with any error:", np.nonzero(d.max(axis=1) > 0)[0].tolist()[:48])

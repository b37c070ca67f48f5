"""
Diagnostic (not a test): how many H1 classes of an audio window are alive at once, at the granularity of the
sweep's 512-edge chunks?  Sizes the class-bit width of the narrow first pass of rips_cloud_kernel.
Uses the CPU oracle (test infrastructure) on the bench's synthetic corpus audio.
    python tests/analysis/class_alive_hist.py [n_rec]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import port                                   # noqa: E402
from tda_eeg_audio_amd import synth                       # noqa: E402

n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 24
aud = synth.corpus_audio(n_rec, 15)
for band in synth.BANDS:
    need, evs, pts, lastr = [], [], [], []
    for r in range(n_rec):
        tau = port.compute_tau(aud[band][r, 0], 125)
        for w in range(15):
            s = aud[band][r, w]
            (h0, h1), P = port.audio_persistence(s, tau)
            if P < 3:
                continue
            pc = port.minmax_normalise(port.takens(s, 3, tau, 2))
            dm = port.cloud_dm(pc).astype(np.float32)
            iu = np.triu_indices(P, 1)
            d = np.sort(dm[iu])
            renc = dm.max(axis=1).min()
            ev = int(np.searchsorted(d, min(renc, 2.0), side="right"))
            fin = h1[np.isfinite(h1[:, 1])]
            b = np.searchsorted(d, fin[:, 0].astype(np.float32), side="left")
            de = np.searchsorted(d, fin[:, 1].astype(np.float32), side="left")
            # exact: classes alive at once at edge granularity (a chunk whose births do not fit is cut short, so the
            # capacity that matters is this one); chunks: how many 512-edge chunks a width of `cap` bits would cut
            ev_t = np.concatenate([b, de]); ev_s = np.concatenate([np.ones_like(b), -np.ones_like(de)])
            o = np.lexsort((ev_s, ev_t))
            mx = int(np.cumsum(ev_s[o]).max()) if len(o) else 0
            need.append(mx); evs.append(ev / len(d)); pts.append(P)
            lastr.append(int(de.max()) if len(de) else 0)
    need = np.array(need); lastr = np.array(lastr)
    q = lambda c: float((need > c).mean())
    print(f"{band:6s} windows {len(need):4d} P {min(pts)}..{max(pts)}  classes at once: mean {need.mean():.1f} max {need.max()} "
          f" >12 {q(12):.3f} >16 {q(16):.3f} >19 {q(19):.3f} >24 {q(24):.3f} >32 {q(32):.4f}   Ev/E mean {np.mean(evs):.2f} max {np.max(evs):.2f}"
          f"  last death rank mean {lastr.mean():.0f} p99 {np.percentile(lastr, 99):.0f}")

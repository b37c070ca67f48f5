#!/usr/bin/env python3
"""Phase-cycle breakdown of the Rips kernels (diagnostic build: make -C tda_eeg_audio_amd/csrc PROFILE=1)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg_prof.so")
from tda_eeg_audio_amd import engine, synth
ctx = _lib.get_ctx(0)
lib = ctx.lib
buf = (C.c_ulonglong * 48)()
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 256      # 256 = one workgroup per CU (latency); 4096: CUs shared
if os.environ.get("TDA_CLASS_WORDS"):                      # e.g. "1,1": the first-pass widths bench.py uses
    ctx.set_class_words(*[int(x) for x in os.environ["TDA_CLASS_WORDS"].split(",")])
W_ = synth.eeg_windows(NW, seed=3)
dist_ = engine.corr_dist_batch(W_, want_corr=False, ctx=ctx)
engine.rips_dm_batch(dist_, ctx=ctx); lib.tda_profile_read(buf, 1)
engine.rips_dm_batch(dist_, ctx=ctx); lib.tda_profile_read(buf, 1)
v = np.array(list(buf), dtype=np.float64); n = v[8]
print(f"eeg: windows={int(n)} E/win={v[9]/n:.0f} cycles/win: keygen={v[0]/n:.0f} sort={v[1]/n:.0f} unpack={v[2]/n:.0f} "
      f"sweep={v[3]/n:.0f} [mask={v[4]/n:.0f} candidates={v[5]/n:.0f} deps={v[6]/n:.0f} scan+kills={v[7]/n:.0f}] kills/win={v[10]/n:.1f} "
      f"list rounds/win={v[11]/n:.1f} entries/round={v[12]/max(v[11],1):.1f} overflow rounds/win={v[13]/n:.2f}\n     phase d: closure={v[16]/n:.0f} list={v[17]/n:.0f} reduce={(v[18]+v[30]+v[31])/n:.0f} (setup {v[30]/n:.0f} + kill loop {v[31]/n:.0f} + rows/barrier {v[18]/n:.0f}) table={v[19]/n:.0f} rest={v[7]/n:.0f}; candidates walked/win={v[20]/n:.0f} [phase b: to barrier={v[21]/n:.0f} walk={v[22]/n:.0f} publish={v[5]/n:.0f}] chunks/win={v[23]/n:.1f} shortened={v[14]/n:.2f} quiet={v[24]/n:.1f} ended early={v[25]/n:.2f}; triangles to test/win={v[27]/n:.0f} on {v[29]/n:.0f} of {v[28]/n:.0f} apparent edges; dependency rounds/win={v[32]/n:.1f} for {v[33]/n:.0f} in-chunk dependent edges in {v[34]/n:.1f} busy chunks; edges skipped to the next candidate/win={v[35]/n:.0f}")
for band in (os.environ.get("TDA_PROFILE_BANDS", "beta,delta").split(",")):
    wins = synth.audio_windows(NW, band, seed=1)
    tau = engine.tau_batch(wins[:1], 125, ctx=ctx)[0]
    engine.takens_rips_batch(wins, tau, ctx=ctx)
    lib.tda_profile_read(buf, 1)
    engine.takens_rips_batch(wins, tau, ctx=ctx)
    lib.tda_profile_read(buf, 1)
    v = np.array(list(buf), dtype=np.float64)
    n = v[8]
    print(f"audio {band} tau={tau}: windows={int(n)} E/win={v[9]/n:.0f} cycles/win: keygen={v[0]/n:.0f} sort={v[1]/n:.0f} "
          f"unpack={v[2]/n:.0f} sweep={v[3]/n:.0f} [mask={v[4]/n:.0f} candidates={v[5]/n:.0f} deps={v[6]/n:.0f} scan+kills={v[7]/n:.0f}] kills/win={v[10]/n:.1f} "
          f"list rounds/win={v[11]/n:.1f} entries/round={v[12]/max(v[11],1):.1f} overflow rounds/win={v[13]/n:.2f}\n     phase d: closure={v[16]/n:.0f} list={v[17]/n:.0f} reduce={(v[18]+v[30]+v[31])/n:.0f} (setup {v[30]/n:.0f} + kill loop {v[31]/n:.0f} + rows/barrier {v[18]/n:.0f}) table={v[19]/n:.0f} rest={v[7]/n:.0f}; candidates walked/win={v[20]/n:.0f} [phase b: to barrier={v[21]/n:.0f} walk={v[22]/n:.0f} publish={v[5]/n:.0f}] chunks/win={v[23]/n:.1f} shortened={v[14]/n:.2f} quiet={v[24]/n:.1f} ended early={v[25]/n:.2f}; triangles to test/win={v[27]/n:.0f} on {v[29]/n:.0f} of {v[28]/n:.0f} apparent edges; dependency rounds/win={v[32]/n:.1f} for {v[33]/n:.0f} in-chunk dependent edges in {v[34]/n:.1f} busy chunks; edges skipped to the next candidate/win={v[35]/n:.0f}"
          f"\n     ranks whose class vectors are needed/win={v[39]/n:.0f} of Ev={v[40]/n:.0f}; windows needing > 4096 / 4608 / 5120: {v[36]/n:.4f} / {v[37]/n:.4f} / {v[38]/n:.4f}")

# ---- Wasserstein phases on pipeline diagrams ----
W = synth.eeg_windows(256, seed=3)
aw = synth.audio_windows(256, "beta", seed=4)
dist = engine.corr_dist_batch(W, want_corr=False, ctx=ctx)
e0, ec0, e1, ec1, est = engine.rips_dm_batch(dist, ctx=ctx, raw=True)
a0, ac0, a1, ac1, npts, ast = engine.takens_rips_batch(aw, 3, ctx=ctx, raw=True)
for name, (x, cx, y, cy) in {"H0": (e0, ec0, a0, ac0), "H1": (e1, ec1, a1, ac1)}.items():
    engine.wasserstein_batch(x, cx, y, cy, ctx=ctx)
    lib.tda_profile_read_ws(buf, 1)
    engine.wasserstein_batch(x, cx, y, cy, ctx=ctx)
    lib.tda_profile_read_ws(buf, 1)
    v = np.array(list(buf), dtype=np.float64); n = v[4]
    print(f"wasserstein {name}: pairs={int(n)} rows={v[5]/n:.1f} cols={v[6]/n:.1f} cycles/pair: setup={v[0]/n:.0f} "
          f"solve={v[1]/n:.0f} total={v[2]/n:.0f}; dijkstra steps/pair={v[3]/n:.1f} -> {v[1]/max(v[3],1):.0f} cycles/step")

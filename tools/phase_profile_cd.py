#!/usr/bin/env python3
"""Phase-cycle breakdown of corr_dist_kernel (diagnostic build: make -C tda_eeg_audio_amd/csrc PROFILE=1)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg_prof.so")
import torch
from tda_eeg_audio_amd import engine
ctx = _lib.get_ctx(0)
buf = (C.c_ulonglong * 16)()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11360
w = torch.randn((n, 47, 250), dtype=torch.float64, device="cuda")
d = torch.empty((n, 47, 47), dtype=torch.float64, device="cuda")
for _ in range(2):
    engine.corr_dist_dev(w, d, ctx=ctx); torch.cuda.synchronize()
    ctx.lib.tda_profile_read_cd(buf, 1)
v = np.array(list(buf), dtype=np.float64); k = v[14]
print(f"corr_dist: windows={int(k)}; ticks (1/2.4 GHz) per window, seen by wave 0")
print("  pass A: issue fetch=%.0f stash(+wait for HBM)=%.0f sync=%.0f sums=%.0f sync=%.0f mean=%.0f" % tuple(v[[9, 10, 11, 12, 13, 0]] / k))
print("  pass B: other=%.0f stash=%.0f sync=%.0f mfma=%.0f sync=%.0f tail=%.0f" % tuple(v[[3, 4, 5, 6, 7, 1]] / k))
print("  epilogue=%.0f   total=%.0f" % (v[2] / k, v[:14].sum() / k))

"""The recordings_from_host leg of bench.py alone (process_recording whole, from pinned host memory).
python tools/recordings_bench.py [n_rec] [shard]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from tda_eeg_audio_amd import _lib
torch.cuda.set_device(0)
ctx = _lib.get_ctx(0)
ctx.set_class_words(1, 1)            # the first-pass widths bench.py uses
n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 708
shard = int(sys.argv[2]) if len(sys.argv) > 2 else 236
r = bench.recordings_leg(ctx, torch.device("cuda", 0), n_rec=n_rec, shard=shard)
print(json.dumps({k: v for k, v in r.items() if k not in ("stage", "sample")}))

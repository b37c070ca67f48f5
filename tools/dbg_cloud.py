import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import port, brute
from tda_eeg_audio_amd import engine, _lib
ctx=_lib.get_ctx(0)
k = 60
ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
a = np.stack([np.cos(ang), np.sin(ang), np.zeros(k)], 1)
b = np.stack([np.cos(ang + np.pi / k), np.sin(ang + np.pi / k), np.full(k, 0.9)], 1)
pc = np.concatenate([a, b])
o = port.rips_f32(port.cloud_dm(pc).astype(np.float32), thresh=0.95)
print("oracle h1", len(o[1]), "h0", len(o[0]))
for wc in (1,2):
    ctx.set_class_words(2, wc)
    h0,h1,st = engine.cloud_rips_batch(pc[None], normalise=False, thresh=0.95, h1_cap=1024, ctx=ctx)
    print("words_cloud",wc,"status",st,"h1",len(h1[0]),"h0",len(h0[0]), "ok", np.array_equal(brute.sort_rows(h1[0]),brute.sort_rows(o[1])))
# same via dm path
dm = port.cloud_dm(pc)
for wd in (1,2,4):
    ctx.set_class_words(wd, 1)
    h0,h1,st = engine.rips_dm_batch(dm[None], thresh=0.95, h1_cap=1024, symmetrise=False, ctx=ctx)
    print("dm words",wd,"status",st,"h1",len(h1[0]), "ok", np.array_equal(brute.sort_rows(h1[0]),brute.sort_rows(o[1])))

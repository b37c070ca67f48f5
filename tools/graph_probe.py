"""Does one step capture into a HIP graph (torch.cuda.CUDAGraph), and what does replay cost?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tda_eeg_audio_amd import _lib, pipeline, synth

dev = torch.device("cuda", 0)
ctx = _lib.get_ctx(0)
n_win, wpr = 710, 15
seg = np.array(list(range(0, n_win, wpr)) + [n_win], np.int32)
eeg_t = torch.from_numpy(synth.eeg_windows(n_win, seed=42, windows_per_recording=wpr)).to(dev)
aud_t = torch.from_numpy(synth.audio_windows(n_win, "beta", seed=4242)).to(dev)
LANES = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wss = [pipeline.Workspace(n_win, seg, dev) for _ in range(LANES)]
streams = [torch.cuda.Stream(device=dev) for _ in range(LANES)]
# eager reference + warm-up (lazy inits must not happen during capture)
for ws, st in zip(wss, streams):
    with torch.cuda.stream(st):
        for _ in range(2):
            pipeline.run_step(eeg_t, aud_t, ws, ctx=ctx)
torch.cuda.synchronize()
ref = wss[0].result.clone()
graphs = []
for ws, st in zip(wss, streams):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        pipeline.run_step(eeg_t, aud_t, ws, ctx=ctx)
    graphs.append(g)
torch.cuda.synchronize()
print("captured", len(graphs), "graphs")
wss[0].result.zero_()
graphs[0].replay()
torch.cuda.synchronize()
print("replay result identical to eager:", bool(torch.equal(wss[0].result, ref)))
for steps in (60, 300):
    t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(streams[k % LANES]):
            graphs[k % LANES].replay()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"graph replay, {LANES} lanes, {steps} steps: {dt * 1e3:.4f} ms/step -> {n_win / dt:.0f} windows/s (host enqueue {t_enq / steps * 1e3:.4f} ms/step)")

"""Does the EEG chain hide behind the audio Rips kernel once both fit on a CU together?
Experiment builds: libtdaeeg_exp4.so / libtdaeeg_exp5.so = launch bound of 4 / 5 waves per
SIMD for rips_cloud_kernel (124 / 96 VGPRs), -DTDA_EXPERIMENT so that TDA_EXP_PMAX shrinks the LDS layout.
Audio clouds are made small (tau fixed at 45 -> 80 points, 46 KB of LDS) so that two audio workgroups and an EEG
workgroup fit in LDS; only the register budget then decides about co-residency.
usage: TDA_EXP_PMAX=80 python tools/coresidency_probe.py exp4|exp5"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tda_eeg_audio_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tda_eeg_audio_amd", f"libtdaeeg_{sys.argv[1]}.so")
import numpy as np, torch
from tda_eeg_audio_amd import engine, pipeline, synth
dev = torch.device("cuda", 0)
ctx = _lib.get_ctx(0)
ctx.set_class_words(1, 1)
n_win, wpr = 710, 15
seg = np.array(list(range(0, n_win, wpr)) + [n_win], np.int32)
eeg_t = torch.from_numpy(synth.eeg_windows(n_win, seed=42, windows_per_recording=wpr)).to(dev)
aud_t = torch.from_numpy(synth.audio_windows(n_win, "delta", seed=4242)).to(dev)
TAU = int(os.environ.get("TDA_EXP_TAU", "45"))
REAL = {k: getattr(engine, k) for k in dir(engine) if k.endswith("_dev")}

def fixed_tau(win_t, seg_off_t, max_lag=None, tau_seg_t=None, tau_win_t=None, ctx=None):
    tau_seg_t.fill_(TAU)
    if tau_win_t is not None:
        tau_win_t.fill_(TAU)
    return tau_seg_t

def run(label, stub, steps=150):
    for k, v in REAL.items():
        setattr(engine, k, v)
    engine.tau_segments_dev = fixed_tau
    for k in stub:
        setattr(engine, k, lambda *a, **kw: None)
    L = pipeline.Lanes(3, n_win, seg, dev, graph=True)
    for _ in range(9):
        L.submit(eeg_t, aud_t, ctx=ctx, sync_inputs=False)
    L.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        L.submit(eeg_t, aud_t, ctx=ctx, sync_inputs=False)
    L.drain(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = L.ws[0].aud.status.cpu().numpy()
    print(f"{sys.argv[1]} {label:40s} {dt * 1e3:.4f} ms/step   (audio status bits seen: {sorted(set(st.tolist()))})", flush=True)

ALL_BUT_AUDIO = ["corr_dist_dev", "rips_dm_dev", "wasserstein_dev", "features_dev", "recording_rows_dev"]
for rep in range(2):
    run("full step", [])
    run("audio Rips only", ALL_BUT_AUDIO)
    run("EEG chain only (corr + Rips)", ["takens_rips_dev", "wasserstein_dev", "features_dev", "recording_rows_dev"])

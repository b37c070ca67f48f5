"""Stand-alone timing of single kernels of the hot path (HIP events on torch's current stream).

    python tools/kernel_bench.py corr [n_win]      corr_dist_kernel, stacked windows
    python tools/kernel_bench.py sliding [n_samp]  corr_dist_kernel, sliding windows over one recording
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tda_eeg_audio_amd import engine


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "corr"
    g = torch.Generator(device="cuda").manual_seed(0)
    if what == "corr":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 11360
        w = torch.randn((n, 47, 250), dtype=torch.float64, device="cuda", generator=g)
        d = torch.empty((n, 47, 47), dtype=torch.float64, device="cuda")
        ms = timeit(lambda: engine.corr_dist_dev(w, d))
        gb = n * (47 * 250 + 47 * 47) * 8 / 1e9
        print(f"corr_dist  n_win={n}  {ms:.4f} ms  {ms * 1e3 / n:.4f} us/window  {gb / ms * 1e3:.1f} GB/s algorithmic")
    elif what == "sliding":
        L = int(sys.argv[2]) if len(sys.argv) > 2 else 700000
        s = torch.randn((47, L), dtype=torch.float64, device="cuda", generator=g)
        n = (L - 250) // 62 + 1
        d = torch.empty((n, 47, 47), dtype=torch.float64, device="cuda")
        ms = timeit(lambda: engine.corr_dist_sliding_dev(s, 250, 62, d))
        gb = (47 * L + n * 47 * 47) * 8 / 1e9
        print(f"corr_dist_sliding  n_win={n}  {ms:.4f} ms  {ms * 1e3 / n:.4f} us/window  {gb / ms * 1e3:.1f} GB/s algorithmic")


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "audio"):
    main()


def audio_only(n=710, lanes=3, steps=60):
    """Upper bound probe: only the audio Rips stage, `lanes` batches in flight."""
    import numpy as np
    from tda_eeg_audio_amd import synth
    aud = torch.from_numpy(synth.audio_windows(n, "beta", seed=4242)).cuda()
    tau = torch.full((n,), 3, dtype=torch.int32, device="cuda")
    outs = [engine.DeviceDiagrams(n, 128, engine.DEFAULT_H1_CAP, aud.device) for _ in range(lanes)]
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    def run(k):
        with torch.cuda.stream(streams[k % lanes]):
            engine.takens_rips_dev(aud, tau, outs[k % lanes])
    for k in range(6):
        run(k)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for k in range(steps):
        run(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"audio Rips only: n={n} lanes={lanes}: {dt * 1e3:.4f} ms per batch -> {n / dt:.0f} windows/s")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "audio":
    for L in (1, 2, 3):
        audio_only(lanes=L)

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


if __name__ == "__main__":
    main()

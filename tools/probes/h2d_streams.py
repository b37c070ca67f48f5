"""Host-to-device bandwidth from pinned memory with 1 / 2 / 4 copies in flight (torch streams -> SDMA engines), and with
the copy done by a kernel reading the pinned host buffer directly (torch .copy_ from a pinned tensor mapped into the GPU's
address space is still an SDMA copy; the kernel path is tda's own upload if it pays).   python tools/probes/h2d_streams.py"""
import time
import torch

dev = torch.device("cuda", 0)
GB = 1 << 30
n = GB // 8
host = torch.empty(2 * n, dtype=torch.float64).pin_memory()
host.normal_()
dst = torch.empty(2 * n, dtype=torch.float64, device=dev)
for k in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
    parts = torch.chunk(torch.arange(2 * n), k)
    bounds = [(int(p[0]), int(p[-1]) + 1) for p in parts]
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s, (a, b) in zip(streams, bounds):
            with torch.cuda.stream(s):
                dst[a:b].copy_(host[a:b], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{k} copies in flight: {2 * GB / best / 1e9:6.1f} GB/s  (2 GiB in {best * 1e3:.1f} ms)")
# small pieces, as the recordings pass uploads them (236 recordings x 47 x 4606 f64 = 409 MB per shard)
for mb in (16, 64, 256):
    m = mb * (1 << 20) // 8
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(0, 2 * n - m + 1, m):
            dst[i:i + m].copy_(host[i:i + m], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"one stream, pieces of {mb} MB: {(2 * n // m) * m * 8 / best / 1e9:6.1f} GB/s")

"""Reference points for HBM streaming on this box: torch copy of the same byte counts as corr_dist."""
import torch, sys
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11360
src = torch.randn((n, 47, 250), dtype=torch.float64, device="cuda")
dst = torch.empty_like(src)
red = torch.empty((n, 47, 47), dtype=torch.float64, device="cuda")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = timeit(lambda: dst.copy_(src))
print(f"copy {src.numel()*8/1e9:.2f} GB: {ms:.4f} ms -> read+write {2*src.numel()*8/ms/1e6:.0f} GB/s")
ms = timeit(lambda: torch.sum(src, dim=2, out=red[:, :, 0]))
print(f"row-sum (read only): {ms:.4f} ms -> {src.numel()*8/ms/1e6:.0f} GB/s")

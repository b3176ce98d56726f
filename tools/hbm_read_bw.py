#!/usr/bin/env python3
"""Measured streaming-read ceiling of this GPU (for the roofline discussion in
DESIGN.md): torch reductions / copies over buffers far larger than the 256 MB
Infinity Cache.  Prints GB/s."""
import torch

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

n = 4 << 30  # 4 Gi int32 = 16 GiB
x = torch.ones(n, dtype=torch.int32, device="cuda")
t = timeit(lambda: x.sum())
print(f"read  (sum int32, {n*4/2**30:.0f} GiB): {n*4/t/1e9:.0f} GB/s")
xf = x.view(torch.float32)
t = timeit(lambda: xf.max())
print(f"read  (max f32): {n*4/t/1e9:.0f} GB/s")
y = torch.empty_like(x[: n // 2])
t = timeit(lambda: y.copy_(x[: n // 2]))
print(f"copy  (8 GiB -> 8 GiB): {n*4/t/1e9:.0f} GB/s (read+write)")

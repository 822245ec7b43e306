#!/usr/bin/env python3
"""Device-to-device copy rate of the GPU, as a check on the HBM peak the
roofline is priced against: copies a buffer far larger than the 256 MB
Infinity Cache and reports read + write bytes per second."""
import torch

n = 4 << 30
src = torch.empty(n, dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
src.fill_(1)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    dst.copy_(src)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"copy of {n >> 30} GiB: {ms:.3f} ms, {2 * n / ms / 1e9:.2f} TB/s read + write")
a = torch.empty(n // 4, dtype=torch.float32, device="cuda").fill_(1.0)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    s = a.sum()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"read-only sum of {n >> 30} GiB: {ms:.3f} ms, {n / ms / 1e9:.2f} TB/s")

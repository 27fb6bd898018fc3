#!/usr/bin/env python3
"""Calibration: device read-only / write-only / copy rates with torch's own kernels (GB/s of bytes touched)."""
import torch
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    xi = x.view(torch.int32)
    r = t(lambda: xi.sum())
    r2 = t(lambda: torch.max(x))
    w = t(lambda: y.zero_())
    c = t(lambda: y.copy_(x))
    b = n * 4 / 1e9
    print(f"{mb:5d} MiB: read(sum) {b / r:7.0f}  read(max) {b / r2:7.0f}  write(zero) {b / w:7.0f}  copy {2 * b / c:7.0f} GB/s", flush=True)
    del x, y, xi

#!/usr/bin/env python3
"""Practical HBM ceilings of this box, for the roofline discussion in DESIGN.md: read-only (sum), write-only (fill), copy, and
read-modify-write (add_), on 1 GiB fp32 tensors, best of 20 after warm-up, torch kernels (vectorised elementwise / reduce)."""
import time
import torch
dev = torch.device("cuda:0")
n = 1 << 28
x = torch.ones(n, device=dev); y = torch.empty_like(x)
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize(); best = min(best, a.elapsed_time(b))
    return best
gb = n * 4 / 1e9
for name, f, vol in (("read-only  (sum)", lambda: x.sum(), gb), ("write-only (fill_)", lambda: y.fill_(2.0), gb),
                     ("copy       (copy_)", lambda: y.copy_(x), 2 * gb), ("read+write (add_)", lambda: y.add_(1.0), 2 * gb)):
    ms = t(f)
    print(f"{name:20s} {ms:8.3f} ms  {vol / ms:8.2f} TB/s" .replace("TB/s", "GB/ms = TB/s"))

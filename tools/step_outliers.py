#!/usr/bin/env python3
"""Which step of a long run is slow, in which stage, and what did torch's allocator do in that step?   usage: tools/step_outliers.py [workload] [steps]
Per step: the six stage times (stp_timing_history), the step interval, and the deltas of torch's allocator counters (segments allocated / freed =
hipMalloc / hipFree calls, alloc retries) sampled on the host after every step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stopthepop-rasterization_amd"))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import torch
import bench
from diff_gaussian_rasterization import _C
name = args[0] if args else "C5"
steps = int(args[1]) if len(args) > 1 else 200
dev = torch.device("cuda:0")
wl = bench.Workload(name, "full", dev)
for _ in range(10):
    wl.step()
torch.cuda.synchronize(dev)
keys = ("segment.all.allocated", "segment.all.freed", "num_alloc_retries", "allocation.all.allocated")
stat = lambda: [torch.cuda.memory_stats(dev).get(k, 0) for k in keys]
marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
for m in marks:
    m.record()
torch.cuda.synchronize(dev)
_C.timing_enable(True)
prev = stat(); deltas = []
marks[0].record()
for i in range(steps):
    wl.step()
    marks[i + 1].record()
    cur = stat(); deltas.append([c - p for c, p in zip(cur, prev)]); prev = cur
torch.cuda.synchronize(dev)
hist = _C.timing_history(dev, capacity=steps)
_C.timing_enable(False)
seq = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
med = sorted(seq)[steps // 2]
print(f"{name}: {steps} steps, median {med:.3f} ms, max {max(seq):.3f} ms, pool: {_C._native().pooled_sizes(0)}")
smed = {k: sorted(h[k] for h in hist)[steps // 2] for k in hist[0]}
print("median stages:", {k: round(v, 3) for k, v in smed.items()})
for i in range(steps):
    slow = {k: round(v, 3) for k, v in hist[i].items() if v > 1.5 * smed[k] + 0.05}
    if seq[i] > 1.15 * med or slow or deltas[i][0] or deltas[i][1] or deltas[i][2]:
        print(f"step {i}: {seq[i]:.3f} ms  slow stages {slow}  outside {seq[i] - sum(hist[i].values()):.3f}  allocator: segments +{deltas[i][0]} -{deltas[i][1]} retries +{deltas[i][2]}")

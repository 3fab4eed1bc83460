#!/usr/bin/env python3
"""Per-kernel lines of a rocprofv3 kernel_stats.csv whose name matches a regex: calls, average and total microseconds.
   usage: kstats.py <kernel_stats.csv> [regex]"""
import csv, re, sys
rx = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
for r in csv.DictReader(open(sys.argv[1])):
    if rx.search(r["Name"]):
        name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rp::", r["Name"])
        m = re.search(r"(onesweep_\w+|os_\w+_kernel|fillBufferAligned|duplicate_kernel|tile_\w+_kernel)", name)
        print(f"   {(m.group(1) if m else name[:60]):40s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1000:8.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")

#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel (mean per dispatch).  Usage:
   python profiles/pmc_summary.py <dir with *_counter_collection.csv> > summary.txt"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"]
        if "render_hier" not in name and "stp::" not in name:
            continue
        short = name.replace("void ", "").replace("stp::(anonymous namespace)::", "").replace("(stp::RenderArgs)", "")
        short = short.split("(stp::")[0]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[short]["_VGPR"].append(float(r.get("VGPR_Count", 0) or 0))
        acc[short]["_LDS"].append(float(r.get("LDS_Block_Size", 0) or 0))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:28s} mean/dispatch = {sum(v)/len(v):16.1f}   (n={len(v)})")

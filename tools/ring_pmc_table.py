#!/usr/bin/env python3
"""Per-trial means of every counter of one rocprofv3 --pmc pass over tools/ring_levels.py: the named kernel's dispatches in order, grouped into the
probe's trials, beside the Render time the probe reported for the trial.  usage: ring_pmc_table.py <pmc dir> <levels.txt> <kernel substring>"""
import csv, glob, json, sys, collections
d, levels, kern = sys.argv[1], sys.argv[2], sys.argv[3]
trials = [json.loads(l) for l in open(levels) if l.startswith("{")]
per = collections.defaultdict(dict)   # dispatch id -> counter -> value
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if kern in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(per)
if not ids or not trials:
    print("no dispatches / trials", len(ids), len(trials)); sys.exit(0)
n = len(ids) // len(trials)
names = sorted({c for v in per.values() for c in v})
print("trial Render_ms " + " ".join(names))
for t, tr in enumerate(trials):
    grp = ids[t * n:(t + 1) * n][1:]   # (first dispatch of a trial: cold)
    row = [sum(per[i].get(c, 0.0) for i in grp) / max(len(grp), 1) for c in names]
    print(t, tr["Render"], " ".join(f"{v:.4g}" for v in row))

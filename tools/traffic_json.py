#!/usr/bin/env python3
"""Merge the PMC passes of tools/profile.sh into profiles/traffic.json (what bench.py quotes as roofline.traffic / valu).
   usage: python tools/traffic_json.py <workload label, e.g. C2-full> <dir with kernel_stats.csv pmc1.txt pmc4.txt pmc5.txt> [out.json]
Units and corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB per dispatch; on
gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is taken as is (uncalibrated).  The two
counters come from separate passes (they do not fit one).  Pass 1 supplies the SQ instruction counters, kernel_stats.csv the
average launch duration under the profiler.  The file is stamped with the sha256 of csrc/ (bench.csrc_sha256): bench.py
refuses to quote a profile taken on other kernel sources."""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
label, d = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "traffic.json")


def parse(path, counters):
    res, cur = {}, None
    if not os.path.exists(path):
        return res
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch =\s+([0-9.]+)", line)
            if m and m.group(1) in counters and cur:
                res.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    return res


def short(name):  # same shortening as profiles/pmc_summary.py
    s = name.replace("void ", "").replace("stp::(anonymous namespace)::", "").replace("(stp::RenderArgs)", "")
    return s.split("(stp::")[0]


SQ = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_INSTS_LDS")
fetch = parse(os.path.join(d, "pmc4.txt"), ("FETCH_SIZE",))
write = parse(os.path.join(d, "pmc5.txt"), ("WRITE_SIZE",))
sq = parse(os.path.join(d, "pmc1.txt"), SQ)
avg_ms = {}
ks = os.path.join(d, "kernel_stats.csv")
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        avg_ms[short(r["Name"])] = float(r["AverageNs"]) * 1e-6
try:
    allj = json.load(open(out))
except (OSError, ValueError):
    allj = {}
import bench  # noqa: E402  (csrc_sha256 only)
stamp = bench.csrc_sha256()
if allj.get("_csrc_sha256") != stamp:   # profiles of other kernel sources do not mix
    allj = {}
entry = {}
for k in sorted(set(fetch) | set(write) | set(sq)):
    f_b = 2.0 * 1024.0 * fetch.get(k, {}).get("FETCH_SIZE", 0.0)
    w_b = 1024.0 * write.get(k, {}).get("WRITE_SIZE", 0.0)
    e = {"fetch_bytes": int(f_b), "write_bytes": int(w_b), "hbm_bytes_per_launch": int(f_b + w_b)}
    e.update(sq.get(k, {}))
    if k in avg_ms:
        e["avg_ms_at_profile"] = round(avg_ms[k], 5)
    entry[k] = e
allj[label] = entry
allj["_csrc_sha256"] = stamp
allj.setdefault("_dirs", {})[label] = os.path.basename(os.path.normpath(d))
allj["_taken"] = "rocprofv3 PMC passes " + ", ".join(f"profiles/{v} ({k})" for k, v in sorted(allj["_dirs"].items())) + f"; csrc sha256 {stamp[:12]}"
allj["_note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean per dispatch; FETCH_SIZE x2 (gfx950), KiB -> bytes; SQ_* from pass 1"
json.dump(allj, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, "with", len(entry), "kernels for", label)

#!/usr/bin/env python3
"""Merge the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh into profiles/traffic.json.
   usage: python tools/traffic_json.py <workload label, e.g. C2-full> <dir with pmc4.txt pmc5.txt> [out.json]
Units and corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports both counters in KiB per dispatch;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is taken as is
(uncalibrated).  The two counters come from separate passes (they do not fit one)."""
import json, os, re, sys

label, d = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")


def parse(path, counter):
    res, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch =\s+([0-9.]+)", line)
            if m and m.group(1) == counter and cur:
                res[cur] = float(m.group(2))
    return res


fetch = parse(os.path.join(d, "pmc4.txt"), "FETCH_SIZE")
write = parse(os.path.join(d, "pmc5.txt"), "WRITE_SIZE")
try:
    allj = json.load(open(out))
except (OSError, ValueError):
    allj = {}
entry = {}
for k in sorted(set(fetch) | set(write)):
    f_b = 2.0 * 1024.0 * fetch.get(k, 0.0)
    w_b = 1024.0 * write.get(k, 0.0)
    entry[k] = {"fetch_bytes": int(f_b), "write_bytes": int(w_b), "hbm_bytes_per_launch": int(f_b + w_b)}
allj[label] = entry
allj["_note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean per dispatch; FETCH_SIZE x2 (gfx950), KiB -> bytes"
json.dump(allj, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, "with", len(entry), "kernels for", label)

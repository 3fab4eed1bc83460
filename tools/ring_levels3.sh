#!/bin/bash
# round 3: the blend log's depth (adaptive: changes with the history of forwards) against the level
out=gpurun_out/${1:-ring3}; mkdir -p $out
python tools/ring_levels.py --workload C3 --trials 10 --steps 6 --tag adaptive 2>/dev/null > $out/adaptive.txt
for d in 160 176 192 208 224 240 256; do STP_LOG_DEPTH=$d python tools/ring_levels.py --workload C3 --trials 3 --steps 6 --tag depth$d 2>/dev/null; done > $out/fixed.txt
python - <<'P'
import json
for f in ("adaptive","fixed"):
    for l in open("gpurun_out/ring3/%s.txt" % f):
        r=json.loads(l); print(r["tag"], r["trial"], r["Render"], r["Sort"], "depth", r["log_depth"], "cap", r["bin_cap"], "R", r["R"])
P

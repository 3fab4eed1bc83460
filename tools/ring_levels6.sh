#!/bin/bash
out=gpurun_out/${1:-ring6}; mkdir -p $out
python tools/ring_levels.py --workload C3 --trials 10 --steps 6 --tag train 2>/dev/null > $out/evaltrain.txt
python tools/ring_levels.py --workload C3 --trials 10 --steps 6 --eval --tag eval 2>/dev/null >> $out/evaltrain.txt
python tools/ring_levels.py --workload C5 --trials 8 --steps 6 --tag c5train 2>/dev/null >> $out/evaltrain.txt
python - <<'P'
import json
for l in open("gpurun_out/ring6/evaltrain.txt"):
    r=json.loads(l); print(r["tag"], r["trial"], r["Render"], r["Sort"], r["log_depth"])
P

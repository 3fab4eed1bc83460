#!/bin/bash
out=gpurun_out/${1:-ring4}; mkdir -p $out
for i in 1 2; do python tools/ring_levels.py --workload C3 --trials 8 --steps 6 --slab 3.5 --keep-cache --tag slab$i 2>/dev/null; done > $out/slab.txt
for i in 1 2; do python tools/ring_levels.py --workload C3 --trials 8 --steps 6 --slab 6 --keep-cache --tag slab6_$i 2>/dev/null; done >> $out/slab.txt
python tools/ring_levels.py --workload C3 --trials 8 --steps 6 --keep-cache --tag keep 2>/dev/null >> $out/slab.txt
python tools/ring_levels.py --workload C3 --trials 8 --steps 6 --slab 3.5 --tag slabfree 2>/dev/null >> $out/slab.txt
python - <<'P'
import json
for l in open("gpurun_out/ring4/slab.txt"):
    r=json.loads(l); print(r["tag"], r["trial"], r["Render"], r["Sort"], {k:v[0][-9:] for k,v in r["ptrs"].items()})
P

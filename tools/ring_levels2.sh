#!/bin/bash
# round 2 of the ring-level experiments: many re-allocations per process (level vs a plain read over the same buffers), then the SoA skew
out=gpurun_out/${1:-ring2}; mkdir -p $out
for i in 1 2 3; do python tools/ring_levels.py --workload C3 --trials 10 --steps 8 --tag r$i 2>/dev/null; done > $out/c3_realloc.txt
for skew in 256 4096 4352 69632; do
  for i in 1 2; do STP_CARVE_SKEW=$skew python tools/ring_levels.py --workload C3 --trials 6 --steps 8 --tag skew${skew}_$i 2>/dev/null; done
done > $out/c3_skew.txt
python - <<'P'
import json,sys,collections
for f in ("c3_realloc","c3_skew"):
    rows=[json.loads(l) for l in open(f"gpurun_out/%s/%s.txt" % (sys.argv[1] if len(sys.argv)>1 else "ring2", f))]
    by=collections.defaultdict(list)
    for r in rows: by[r["tag"].rsplit("_",1)[0] if "skew" in r["tag"] else r["tag"]].append((r["Render"], r["Sort"], r.get("binning_read_GBps"), r.get("image_read_GBps")))
    for k,v in by.items(): print(k, v)
P

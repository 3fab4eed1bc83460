#!/bin/bash
# placement sweep: same physical memory (torch's cache keeps it), the image / binning / geometry buffer moved by an offset inside its allocation
out=gpurun_out/${1:-ring5}; mkdir -p $out
K="--workload C3 --steps 6 --keep-cache"
python tools/ring_levels.py $K --tag img_small --img-offsets 0,256,1024,4096,16384,65536,262144,1048576,0 2>/dev/null > $out/sweep.txt
python tools/ring_levels.py $K --tag img_big --img-offsets 0,2097152,4194304,8388608,16777216,33554432,50331648,0x9d800,0 2>/dev/null >> $out/sweep.txt
python tools/ring_levels.py $K --tag bin_small --bin-offsets 0,256,1024,4096,16384,65536,262144,1048576,0 2>/dev/null >> $out/sweep.txt
python tools/ring_levels.py $K --tag bin_big --bin-offsets 0,2097152,4194304,8388608,16777216,33554432,50331648,0x9d800,0 2>/dev/null >> $out/sweep.txt
python tools/ring_levels.py $K --tag geom --geom-offsets 0,4096,65536,1048576,2097152,16777216,33554432,0x9d800,0 2>/dev/null >> $out/sweep.txt
python - <<'P'
import json
for l in open("gpurun_out/ring5/sweep.txt"):
    r=json.loads(l); print(r["tag"], r["trial"], r["Render"], r["Sort"], r["off"], {k:v[0][-9:] for k,v in r["ptrs"].items()})
P

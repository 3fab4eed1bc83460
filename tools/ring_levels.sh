#!/bin/bash
# The experiments behind "the ring kernel's speed levels" (profiles/EXPERIMENTS.md, round 6; probe: tools/ring_levels.py; counters: tools/ring_pmc.sh).
#   usage (GPU box): tools/ring_levels.sh <processes|realloc|depth|slab|offsets|eval> [out dir under gpurun_out/]
exp=${1:-processes}; out=gpurun_out/${2:-ring_$exp}; mkdir -p $out
P="python tools/ring_levels.py --workload C3"
case $exp in
processes)  # fresh processes x re-allocations: does the level belong to the process?
  for i in 1 2 3 4 5 6; do $P --trials 3 --tag p$i 2>/dev/null; done > $out/c3_processes.txt
  for i in 1 2 3 4; do python tools/ring_levels.py --workload C2 --fwd-bwd --trials 3 --tag c2p$i 2>/dev/null; done > $out/c2_processes.txt ;;
realloc)    # ten re-allocations per process (+ a plain read over the same buffers), then the SoA skew
  for i in 1 2 3; do $P --trials 10 --steps 8 --tag r$i 2>/dev/null; done > $out/c3_realloc.txt
  for skew in 256 4096 4352 69632; do for i in 1 2; do STP_CARVE_SKEW=$skew $P --trials 6 --steps 8 --tag skew${skew}_$i 2>/dev/null; done; done > $out/c3_skew.txt ;;
depth)      # the blend log's depth (adaptive / fixed) against the level
  $P --trials 10 --steps 6 --tag adaptive 2>/dev/null > $out/adaptive.txt
  for d in 160 176 192 208 224 240 256; do STP_LOG_DEPTH=$d $P --trials 3 --steps 6 --tag depth$d 2>/dev/null; done > $out/fixed.txt ;;
slab)       # the same physical memory kept (torch's cache) / one slab split into the three buffers
  for i in 1 2; do $P --trials 8 --steps 6 --slab 3.5 --keep-cache --tag slab$i 2>/dev/null; done > $out/slab.txt
  for i in 1 2; do $P --trials 8 --steps 6 --slab 6 --keep-cache --tag slab6_$i 2>/dev/null; done >> $out/slab.txt
  $P --trials 8 --steps 6 --keep-cache --tag keep 2>/dev/null >> $out/slab.txt
  $P --trials 8 --steps 6 --slab 3.5 --tag slabfree 2>/dev/null >> $out/slab.txt ;;
offsets)    # one buffer at a time moved inside its allocation, physical memory kept
  K="--steps 6 --keep-cache"
  $P $K --tag img_small --img-offsets 0,256,1024,4096,16384,65536,262144,1048576,0 2>/dev/null > $out/sweep.txt
  $P $K --tag img_big --img-offsets 0,2097152,4194304,8388608,16777216,33554432,50331648,0x9d800,0 2>/dev/null >> $out/sweep.txt
  $P $K --tag bin_small --bin-offsets 0,256,1024,4096,16384,65536,262144,1048576,0 2>/dev/null >> $out/sweep.txt
  $P $K --tag bin_big --bin-offsets 0,2097152,4194304,8388608,16777216,33554432,50331648,0x9d800,0 2>/dev/null >> $out/sweep.txt
  $P $K --tag geom --geom-offsets 0,4096,65536,1048576,2097152,16777216,33554432,0x9d800,0 2>/dev/null >> $out/sweep.txt ;;
eval)       # the forward that records no log against the training forward
  $P --trials 10 --steps 6 --tag train 2>/dev/null > $out/evaltrain.txt
  $P --trials 10 --steps 6 --eval --tag eval 2>/dev/null >> $out/evaltrain.txt
  python tools/ring_levels.py --workload C5 --trials 8 --steps 6 --tag c5train 2>/dev/null >> $out/evaltrain.txt ;;
esac
cat $out/*.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['tag'], r['trial'], r['Render'], r['Sort'], 'depth', r.get('log_depth'), 'off', r.get('off'), {k: v[0][-9:] for k, v in r['ptrs'].items()})"

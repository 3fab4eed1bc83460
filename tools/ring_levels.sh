#!/bin/bash
# the ring kernel's speed levels by process and by buffer placement (tools/ring_levels.py); run on the GPU box
out=gpurun_out/${1:-ring}; mkdir -p $out
for i in 1 2 3 4 5 6; do python tools/ring_levels.py --workload C3 --trials 3 --tag p$i 2>/dev/null; done > $out/c3_processes.txt
for i in 1 2; do python tools/ring_levels.py --workload C3 --pads 0,4096,65536,1048576,2097152,33554432,0,12288 --tag pad$i 2>/dev/null; done > $out/c3_pads.txt
for i in 1 2 3 4; do python tools/ring_levels.py --workload C2 --fwd-bwd --trials 3 --tag c2p$i 2>/dev/null; done > $out/c2_processes.txt
cat $out/c3_processes.txt $out/c3_pads.txt $out/c2_processes.txt | cut -c1-400

#!/bin/bash
# One GPU-box call that refreshes everything under profiles/ for a round: bench lines of every workload (no profiler
# attached, taken FIRST), then kernel trace + PMC passes of C2-full and C2-min (all six) and of C3 / C4 / C5 (passes 1, 4, 5: what
# profiles/traffic.json needs), then profiles/traffic.json.
#   usage (on the GPU box, from the repo root): tools/round_profiles.sh r06
set -u
tag=${1:-rXX}
out=gpurun_out/${tag}
mkdir -p $out
B="timeout -k 5 600 python bench.py"
$B > $out/bench_full.json 2> $out/bench_full.err   # (the driver's line: headline + checker legs + other_workloads)
cp bench_detail.json $out/bench_full_detail.json 2>/dev/null
$B --variant min --no-cpu-baseline --no-other-workloads > $out/bench_min.json 2>/dev/null
$B --fwd-only --no-cpu-baseline --no-other-workloads > $out/bench_full_fwd.json 2>/dev/null
$B --workload C1 --steps 500 --warmup 50 --no-cpu-baseline --no-other-workloads > $out/bench_c1.json 2>/dev/null
$B --workload C3 --no-cpu-baseline --no-other-workloads > $out/bench_c3.json 2>/dev/null
$B --workload C4 --no-cpu-baseline --no-other-workloads > $out/bench_c4_fwd.json 2>/dev/null
$B --workload C5 --no-cpu-baseline --no-other-workloads > $out/bench_c5.json 2>/dev/null
$B --workload L1 --steps 10 --no-cpu-baseline --no-other-workloads > $out/bench_l1.json 2>/dev/null
$B --workload C2L --no-cpu-baseline --no-other-workloads > $out/bench_c2l.json 2>/dev/null
$B --workload C2H --no-cpu-baseline --no-other-workloads > $out/bench_c2h.json 2>/dev/null
tools/profile.sh ${tag}_full full > /dev/null 2>&1
tools/profile.sh ${tag}_min min > /dev/null 2>&1
PMC_PASSES="1 4 5" tools/profile.sh ${tag}_c3 full --workload C3 > /dev/null 2>&1
PMC_PASSES="1 4 5" tools/profile.sh ${tag}_c4 full --workload C4 > /dev/null 2>&1
PMC_PASSES="1 4 5" tools/profile.sh ${tag}_c5 full --workload C5 > /dev/null 2>&1
python tools/traffic_json.py C2-full gpurun_out/${tag}_full gpurun_out/${tag}/traffic.json
python tools/traffic_json.py C2-min gpurun_out/${tag}_min gpurun_out/${tag}/traffic.json
python tools/traffic_json.py C3-full gpurun_out/${tag}_c3 gpurun_out/${tag}/traffic.json
python tools/traffic_json.py C4-full gpurun_out/${tag}_c4 gpurun_out/${tag}/traffic.json
python tools/traffic_json.py C5-full gpurun_out/${tag}_c5 gpurun_out/${tag}/traffic.json
for f in $out/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['stage_ms'])" 2>/dev/null; done

#!/bin/bash
# Kernel-trace view of the tile-bit sort on the GPU box: our own onesweep driver against rocprim::radix_sort_pairs (STP_TILE_SORT=rocprim).
#   tools/sortprof.sh > gpurun_out/sortprof.txt      (per-kernel durations mislead here: see stp_binning.hip; the stage A/B is tools/ab_env.sh)
export TMPDIR=/tmp; root=$(pwd); cd /tmp
for v in default rocprim; do
  rm -rf /tmp/sp_$v; STP_TILE_SORT=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$v -- python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
  f=$(find /tmp/sp_$v -name '*kernel_stats.csv' | head -1); echo "== STP_TILE_SORT=$v"; python3 $root/tools/kstats.py $f "os_|rocprim|fillBuffer|duplicate|sh_color"
done

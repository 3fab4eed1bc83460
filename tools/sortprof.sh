#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); cd /tmp
for v in 0; do
  rm -rf /tmp/sp$v; STP_OS_DEBUG=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp$v -- python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
  f=$(find /tmp/sp$v -name '*kernel_stats.csv' | head -1); echo "== STP_OS_DEBUG=$v"; python3 $root/tools/kstats.py $f "os_|fillBuffer|duplicate"
done
rm -rf /tmp/sp2; STP_TILE_SORT=rocprim rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp2 -- python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
f=$(find /tmp/sp2 -name '*kernel_stats.csv' | head -1); echo "== rocprim"; python3 $root/tools/kstats.py $f "rocprim|fillBuffer|duplicate"

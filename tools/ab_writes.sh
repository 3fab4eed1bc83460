#!/bin/bash
# WRITE_SIZE (one --pmc pass, no trace) + forward time of library builds, training forwards only: tools/ab_writes.sh "<workloads>" <lib...>
W=$1; shift
root=$(pwd); export TMPDIR=/tmp
for w in $W; do for L in "$@"; do
  n=$(basename $L .so); d=/tmp/abw_${w}_$n; rm -rf $d
  (cd /tmp && STP_RASTER_LIB=$(realpath $root/$L) timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d -- python $root/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --train-forward-only > /dev/null 2>&1)
  echo -n "$w $n "; python $root/profiles/pmc_summary.py $d | grep -A3 "render_hier_kernel<4, 8, true, 2\|render_kbuffer_ring_kernel<16, 2" | grep WRITE_SIZE | head -1
done; done

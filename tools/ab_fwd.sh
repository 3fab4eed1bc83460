#!/bin/bash
# alternating A/B of library builds, TRAINING FORWARDS ONLY (recording forward without its backward: for builds whose log is wrong on purpose)
#   tools/ab_fwd.sh "<workloads>" <rounds> <lib...>
W=$1; R=$2; shift 2
for i in $(seq $R); do for w in $W; do for L in "$@"; do
  echo -n "$w $(basename $L) "; STP_RASTER_LIB=$(realpath $L) python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --train-forward-only 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('Sort %.4f Render %.4f' % (s.get('Sort',0), s['Render']))"
done; done; done

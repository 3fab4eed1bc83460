#!/bin/bash
# alternating A/B of library builds over workloads: tools/ab_libs.sh "<workloads>" <rounds> <lib...>   (stage times, 20 steps each)
W=$1; R=$2; shift 2
for i in $(seq $R); do for w in $W; do for L in "$@"; do
  echo -n "$w $(basename $L) "; STP_RASTER_LIB=$(realpath $L) python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('fps %.1f ms %.4f Sort %.4f Render %.4f BwdRender %.4f' % (d['value'], d['ms_per_step'], s.get('Sort',0), s['Render'], s.get('BwdRender',0)))"
done; done; done

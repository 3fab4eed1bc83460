#!/bin/bash
# A/B of library builds on several workloads, alternating: tools/ab_workloads.sh <rounds> "<workload[:variant]> ..." <lib...>
#   e.g. tools/ab_workloads.sh 2 "C2:full C3 C5" libstp_raster.so libstp_raster_ieee.so
R=$1; WL="$2"; shift 2
for i in $(seq $R); do for w in $WL; do for L in "$@"; do
  name=${w%%:*}; var=${w#*:}; [ "$var" = "$w" ] && var=full
  echo -n "$w $(basename $L) "
  STP_RASTER_LIB=$(realpath $L) python bench.py --workload $name --variant $var --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print(' '.join('%s %.4f' % (k, v) for k, v in s.items()), 'total %.4f median %.4f fps %.1f' % (d['ms_per_step'], d['step_ms']['median'], d['value']))"
done; done; done

#!/bin/bash
# A/B of library builds on one box, printing one stage (default Preprocess): tools/ab_stage.sh <stage> <rounds> <lib...>
S=$1; R=$2; shift 2
for i in $(seq $R); do for L in "$@"; do
  STP_RASTER_LIB=$(realpath $L) python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('$(basename $L)', '$S %.4f total %.4f fps %.1f' % (s['$S'], d['ms_per_step'], d['value']))"
done; done

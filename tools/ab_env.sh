#!/bin/bash
# A/B of an environment switch on several workloads, alternating: tools/ab_env.sh <rounds> "<workload[:variant[:steps]]> ..." "<ENV=a>" "<ENV=b>" ...
#   e.g. tools/ab_env.sh 2 "C1:full:500 C2:full C3" "STP_RUN_AHEAD=1" "STP_RUN_AHEAD=0"
R=$1; WL="$2"; shift 2
for i in $(seq $R); do for w in $WL; do for E in "$@"; do
  IFS=: read name var steps <<< "$w"; var=${var:-full}; steps=${steps:-20}
  echo -n "$w $E "
  env $E python bench.py --workload $name --variant $var --steps $steps --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print(' '.join('%s %.4f' % (k, v) for k, v in s.items()), 'sum %.4f total %.4f median %.4f max %.4f fps %.1f' % (sum(s.values()), d['ms_per_step'], d['step_ms']['median'], d['step_ms']['max'], d['value']))"
done; done; done

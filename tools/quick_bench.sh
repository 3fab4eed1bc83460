#!/bin/bash
# stage times of the BASELINE workloads on the current build (no checker legs): tools/quick_bench.sh [workloads...]
for w in ${@:-C2 C3 C5}; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['stage_ms'])"
done

#!/bin/bash
# A/B/... timing of several library builds on the same box, alternating: tools/abn.sh <rounds> <variant> <lib...>
R=$1; V=$2; shift 2
for i in $(seq $R); do for L in "$@"; do echo -n "$V $(basename $L) "; STP_RASTER_LIB=$(realpath $L) python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --variant $V | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('Dup %.4f Sort %.4f Render %.4f BwdRender %.4f total %.4f fps %.1f' % (s.get('Duplicate',0), s.get('Sort',0), s['Render'], s.get('BwdRender',0), d['ms_per_step'], d['value']))"; done; done

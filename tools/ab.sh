#!/bin/bash
# A/B timing of two builds of the library on the same box: alternates them, prints the stage means of every run.
#   usage: tools/ab.sh <libA.so> <libB.so> [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3 2>/dev/null
for i in $(seq $R); do
  for L in "$A" "$B"; do
    STP_RASTER_LIB=$(realpath $L) python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads "$@" | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('$(basename $L)', 'Render %.4f BwdRender %.4f total %.4f fps %.1f' % (s['Render'], s.get('BwdRender',0), d['ms_per_step'], d['value']))"
  done
done

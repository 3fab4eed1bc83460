#!/bin/bash
# the recording k-buffer forward's log stores, priced (C3, training forwards only: the ablated logs are wrong)
for i in 1 2; do for L in stopthepop-rasterization_amd/diff_gaussian_rasterization/libstp_raster.so gpurun_ab/libstp_kb_logcond.so gpurun_ab/libstp_kb_logrow0.so gpurun_ab/libstp_kb_nolog.so; do
  echo -n "$(basename $L) "; STP_RASTER_LIB=$(realpath $L) python bench.py --workload ${1:-C3} --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --train-forward-only 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('Sort %.4f Render %.4f' % (s.get('Sort',0), s['Render']))"; done; done

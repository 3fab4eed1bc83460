#!/bin/bash
# Quick GPU check used while tuning kernels: parity diag (worst gradient error lines) + stage timings of both C2 variants.
mkdir -p gpurun_out
timeout 600 python tests/gpu_diag.py quick > gpurun_out/diag.log 2>&1
grep -h -E "rel=" gpurun_out/diag.log | sort -t= -k4 -g | tail -2
grep -h -E "PSNR|FAIL|Error|error" gpurun_out/diag.log | sort | uniq -c | tail -4
for v in min full; do
  timeout 300 python bench.py --steps 10 --warmup 3 --variant $v --no-cpu-baseline "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['stage_ms'], d['value'])"
done

#!/bin/bash
# FETCH_SIZE calibration on the GPU box: tools/fetch_calib.sh > gpurun_out/fetch_calib.txt
set -u
root=$(pwd); export TMPDIR=/tmp
[ -x tools/fetch_calib.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib.bin tools/fetch_calib.hip
cd /tmp; rm -rf /tmp/fcal
for c in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/fcal/$(echo $c | tr ' ' '_') -- $root/tools/fetch_calib.bin 2>&1 | grep -E "asked|rror" | head -3
  python $root/profiles/pmc_summary.py /tmp/fcal/$(echo $c | tr ' ' '_')
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fcal/trace -- $root/tools/fetch_calib.bin > /dev/null 2>&1
f=$(find /tmp/fcal/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cat "$f"

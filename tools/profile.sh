#!/bin/bash
# Profile one bench variant on the GPU box: kernel trace + separate PMC passes, summaries into gpurun_out/<tag>/.
#   usage: [PMC_PASSES="1 2"] [STP_RASTER_LIB=...] tools/profile.sh <tag> <variant: full|min> [extra bench args]
# Raw rocprofv3 output goes to /tmp (it is large); only the per-kernel summaries are kept.  Every pass runs under its own timeout (a counter set the
# hardware refuses leaves rocprofv3 hanging in its signal handler: round 6 lost 50 GPU-minutes to two such passes).
set -u
tag=$1; variant=$2; shift 2
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
cmd="python $root/bench.py --steps 5 --warmup 2 --variant $variant --no-cpu-baseline --no-other-workloads $*"
cd /tmp
rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
timeout -k 5 ${PASS_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/trace -- $cmd > "$out/trace_bench.log" 2>&1
f=$(find /tmp/prof_$tag/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
i=0
for set in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_LDS_IDX_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  case " ${PMC_PASSES:-1 2 3 4 5 6} " in *" $i "*) ;; *) continue;; esac
  timeout -k 5 ${PASS_TIMEOUT:-300} rocprofv3 --pmc $set --output-format csv -d /tmp/prof_$tag/pmc$i -- $cmd > "$out/pmc${i}_bench.log" 2>&1
  python $root/profiles/pmc_summary.py /tmp/prof_$tag/pmc$i > "$out/pmc$i.txt" 2>&1
done
cd "$root"

#!/bin/bash
# which counter follows the ring kernel's speed level.  The level is a function of the re-allocation ("trial") inside one
# process (tools/ring_levels.py), so ONE profiled process per counter set gives fast and slow dispatches of the same binary side by side.
# (--pmc passes only, never combined with a trace; every pass under its own timeout: a counter set the hardware refuses leaves rocprofv3 hanging.)
out=$(pwd)/gpurun_out/${1:-ringpmc}; mkdir -p $out; root=$(pwd)
export TMPDIR=/tmp; cd /tmp
i=0
for set in \
 "TCP_CLIENT_UTCL1_INFLIGHT_sum TCP_PENDING_STALL_CYCLES_sum" \
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
 "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" \
 "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
 "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
 "GRBM_EA_BUSY GRBM_TC_BUSY" \
 "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
 "SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
 "SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH SQ_IFETCH_LEVEL" \
 "SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_RES_STALL_CSN" \
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" ; do
  i=$((i+1))
  rm -rf /tmp/ringpmc$i
  timeout -k 5 ${PASS_TIMEOUT:-150} rocprofv3 --pmc $set --output-format csv -d /tmp/ringpmc$i -- python $root/tools/ring_levels.py --workload ${WORKLOAD:-C3} --trials ${TRIALS:-10} --steps 3 --tag pmc$i > $out/levels$i.txt 2> $out/err$i.txt
  python $root/tools/ring_pmc_table.py /tmp/ringpmc$i $out/levels$i.txt ${KERNEL:-render_kbuffer_ring_kernel} > $out/table$i.txt 2>&1
  cat $out/table$i.txt
done

#!/bin/bash
# K2 beyond the five standard PMC groups: instruction fetch and branches, queue levels (average latency = LEVEL / INSTS), the scalar
# caches, the texture addresser / L1 path of the LDS-DMA loads, the dispatcher's resource stalls.  One rocprofv3 run per group
# (--kernel-trace only beside --pmc).  -> gpurun_out/r05_k2_pmc_deep.txt
export TMPDIR=/tmp
OUT=gpurun_out/r05_k2_pmc_deep.txt; : > $OUT
ARGS="--no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --steps 5 --warmup 2"
i=0
for grp in $ONLY_TAIL "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" \
           "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SPI_CSN_BUSY SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN"; do
  i=$((i+1)); rm -rf /tmp/pmcd_$i
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcd_$i -- python bench.py $ARGS > /tmp/pmcd_$i.log 2>&1
  echo "## --pmc $grp" >> $OUT
  db=$(find /tmp/pmcd_$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py pmc "$db" | grep -E "^kernel|k_fast_cells|k_describe|k_pyr" >> $OUT; else echo "(no output)"; tail -3 /tmp/pmcd_$i.log >> $OUT; fi
  echo >> $OUT
done
cat $OUT

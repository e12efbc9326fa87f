#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05_k2_pmc_deep2.txt; : > $OUT
ARGS="--no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --steps 5 --warmup 2"
i=0
for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SPI_CSN_BUSY SPI_RA_LDS_CU_FULL_CSN" "SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN"; do
  i=$((i+1)); rm -rf /tmp/pmce_$i
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmce_$i -- python bench.py $ARGS > /tmp/pmce_$i.log 2>&1
  echo "## --pmc $grp  (rc $?)" >> $OUT
  db=$(find /tmp/pmce_$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py pmc "$db" | grep -E "^kernel|k_fast_cells|k_describe|k_pyr" >> $OUT; else echo "(no output)" >> $OUT; tail -2 /tmp/pmce_$i.log >> $OUT; fi
  echo >> $OUT
done
cat $OUT

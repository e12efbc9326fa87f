#!/bin/bash
# VALU / SALU / LDS instruction counts of K2 (block form) with the stages after a given point switched off
# (PGORB_FAST_DBGSKIP: 1 staging only, 2 + first necessary test, 4 + scores, 3 no minThFAST pass, 0 everything)
export TMPDIR=/tmp
for k in ${SKIPS:-2 4 3 0}; do
  rm -rf gpurun_out/pmcskip_$k
  PGORB_FAST_DBGSKIP=$k rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d gpurun_out/pmcskip_$k -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg > gpurun_out/pmcskip_$k.log 2>&1
  db=$(find gpurun_out/pmcskip_$k -name '*.db' | head -1)
  echo "== DBGSKIP=$k  $(grep -o '"fast": [0-9.]*' gpurun_out/pmcskip_$k.log | tail -1)"; python tools/rocpd_summary.py pmc $db | grep -E "kernel|k_fast"
done

#!/bin/bash
export TMPDIR=/tmp
for v in 1; do
  d=gpurun_out/r3pp_$v; rm -rf $d
  PGORB_FAST_PAIRS=$v rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE -d $d -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 0 --no-verify > /dev/null 2>&1
  db=$(find $d -name '*.db' | head -1)
  echo "pairs=$v"; python tools/rocpd_summary.py pmc $db | grep -E "kernel|k_fast"
done

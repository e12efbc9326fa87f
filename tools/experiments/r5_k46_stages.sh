#!/bin/bash
# Where K4-6 time and instructions go: developer builds that stop a keypoint wave behind a stage (-DPGORB_DESC_STOP=n:
# 1 window staged, 2 + moments and atan2, 3 + row pass, 4 + sin / cos; 0 = everything), each timed by the stage events and counted by one PMC pass.
# Run on the GPU box.
export TMPDIR=/tmp
OUT=gpurun_out/r05_k46_stages.txt; : > $OUT
ARGS="--no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg --no-overlap-leg --no-single-frame-leg"
for k in 1 2 3 4 0; do
  touch pilotguru_amd/csrc/describe.hip
  if [ $k = 0 ]; then make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1; else make -C pilotguru_amd/csrc -j8 EXTRA=-DPGORB_DESC_STOP=$k > /dev/null 2>&1; fi
  ms=$(python bench.py $ARGS --steps 20 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["stage_ms_per_step"]["describe"])')
  rm -rf /tmp/pmcs_$k
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pmcs_$k -- python bench.py $ARGS --steps 5 --warmup 2 > /dev/null 2>&1
  echo "STOP=$k  describe stage $ms ms" | tee -a $OUT
  python tools/rocpd_summary.py pmc $(find /tmp/pmcs_$k -name '*.db' | head -1) | grep -E "^kernel|k_describe" | tee -a $OUT
done

#!/bin/bash
# Per-phase wave times of K2 (developer build -DPGORB_FAST_TIMING, tools/experiments/fast_timing.py) for build variants given as
# EXTRA flag sets, e.g.:  r5_k2_timing.sh "" "-DPG_FAST_SCORE_I32=1".  Run on the GPU box; restores the product build at the end.
OUT=gpurun_out/r05_k2_timing.txt; : > $OUT
for v in "$@"; do
  touch pilotguru_amd/csrc/fast.hip
  make -C pilotguru_amd/csrc -j8 EXTRA="-DPGORB_FAST_TIMING $v" > /dev/null 2>&1
  echo "## EXTRA='$v'" | tee -a $OUT
  python tools/experiments/fast_timing.py 2>&1 | tail -12 | tee -a $OUT
done
touch pilotguru_amd/csrc/fast.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1

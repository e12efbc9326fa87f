#!/bin/bash
# A / B of K2 build variants on ONE box (box-to-box spread of K2 is 2-3 %): EXTRA flag sets as arguments, each built and benched twice, interleaved.
OUT=gpurun_out/r05_k2_ab.txt; : > $OUT
ARGS="--no-cpu-baseline --no-verify --sustain-seconds 1 --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --steps 40 --warmup 10"
for rep in 1 2; do
for v in "$@"; do
  touch pilotguru_amd/csrc/fast.hip
  make -C pilotguru_amd/csrc -j8 EXTRA="$v" > /dev/null 2>&1
  ms=$(python bench.py $ARGS 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["stage_ms_per_step"]["fast"], d["value"])')
  echo "EXTRA='$v'  K2 ms, frames/s: $ms" | tee -a $OUT
done
done
touch pilotguru_amd/csrc/fast.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1

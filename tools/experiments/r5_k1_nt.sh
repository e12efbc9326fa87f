#!/bin/bash
# K1 with non-temporal stores (1), non-temporal LDS-DMA source loads (2), both (3): pyramid stage of the bench, same box.
OUT=gpurun_out/r05_k1_nt.txt; : > $OUT
ARGS="--no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --steps 30 --warmup 10"
for v in "" "-DPGORB_PYR_NT=1" "-DPGORB_PYR_NT=2" "-DPGORB_PYR_NT=3" ""; do
  touch pilotguru_amd/csrc/pyramid.hip
  make -C pilotguru_amd/csrc -j8 EXTRA="$v" > /dev/null 2>&1
  python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EXTRA=$v', d['value'], d['stage_ms_per_step'])" | tee -a $OUT
done
touch pilotguru_amd/csrc/pyramid.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1

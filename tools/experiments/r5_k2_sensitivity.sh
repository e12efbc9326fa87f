#!/bin/bash
# What bounds K2?  The launch's time with one more unit of each resource per cell (developer builds, fast.hip PGORB_FAST_PAD_*):
# if the kernel were bound by VALU issue, 64 more VALU instructions per cell (+ 14 %) would cost + 14 %, and so on.
OUT=gpurun_out/r05_k2_sensitivity.txt; : > $OUT
ARGS="--no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --steps 30 --warmup 10"
for v in "" "-DPGORB_FAST_PAD_VALU=64" "-DPGORB_FAST_PAD_VALU=128" "-DPGORB_FAST_PAD_SALU=64" "-DPGORB_FAST_PAD_SALU=128" "-DPGORB_FAST_PAD_LDS=16" "-DPGORB_FAST_PAD_LDS=32" "-DPGORB_FAST_PAD_SLEEP=4" "-DPGORB_FAST_PAD_SLEEP=16" "" $EXTRA_VARIANTS; do
  touch pilotguru_amd/csrc/fast.hip
  make -C pilotguru_amd/csrc -j8 EXTRA="$v" > /dev/null 2>&1
  ms=$(python bench.py $ARGS 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["stage_ms_per_step"]["fast"])')
  echo "EXTRA='$v'  K2 $ms ms" | tee -a $OUT
done
touch pilotguru_amd/csrc/fast.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1

#!/bin/bash
# K4-6 pipelined form (k_describe_pf): is the wait for the previous keypoint's output stores what it loses?  Timing-only build without them.
OUT=gpurun_out/r05_k46_pf_probe.txt; : > $OUT
ARGS="--no-cpu-baseline --no-verify --sustain-seconds 1 --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --steps 40 --warmup 10"
for v in "" "-DPG_DESC_PF_NOSTORE=1"; do
  touch pilotguru_amd/csrc/describe.hip
  make -C pilotguru_amd/csrc -j8 EXTRA="$v" > /dev/null 2>&1
  for k in 1 4; do
    ms=$(PGORB_DESC_KPW=$k python bench.py $ARGS 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["stage_ms_per_step"]["describe"])')
    echo "EXTRA='$v' KPW=$k  describe $ms ms" | tee -a $OUT
  done
done
touch pilotguru_amd/csrc/describe.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1

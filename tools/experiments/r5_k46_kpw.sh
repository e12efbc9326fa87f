#!/bin/bash
# K4-6 keypoints per wave (k_describe<KPW>, PGORB_DESC_KPW): parity on a slice of the suite, then the stage time. Run on the GPU box.
export TMPDIR=/tmp
OUT=gpurun_out/r05_k46_kpw.txt; : > $OUT
ARGS="--no-cpu-baseline --sustain-seconds 0 --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg"
for k in 1 2 4 8 4 1; do
  export PGORB_DESC_KPW=$k
  line=$(python bench.py $ARGS --steps 30 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["stage_ms_per_step"]["describe"], d["value"], d["verified"])')
  echo "KPW=$k  describe ms, frames/s, verified: $line" | tee -a $OUT
done
for k in 4 8 2; do
  PGORB_DESC_KPW=$k python -m pytest tests/test_gpu_parity.py tests/test_bench_shapes.py -x -q -m gpu -k "stages_and_output or every_frame_of_a_ride or driving_scene or clustered or border or batch_device" 2>&1 | tail -2 | tee -a $OUT
done

#!/bin/bash
# K3 in two launches: parity, then stage times with the option on and off (1080p batch 128, 2160p batch 32, one frame)
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "quadtree or deep or roots or stages or errors" > gpurun_out/r4s/pytest.txt 2>&1; tail -5 gpurun_out/r4s/pytest.txt
for sp in 1 0; do
  echo "== quadtree_split $sp"
  PGORB_QT_SPLIT=$sp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-leg --no-single-frame-leg --no-upload-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1080p', round(d['value']), d['verified'], {k: round(v,4) for k,v in d['stage_ms_per_step'].items()})"
  PGORB_QT_SPLIT=$sp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-leg --no-single-frame-leg --no-upload-leg --width 3840 --height 2160 --features 4000 --batch 32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2160p', round(d['value']), d['verified'], {k: round(v,4) for k,v in d['stage_ms_per_step'].items()})"
  PGORB_QT_SPLIT=$sp timeout 300 python tools/single_frame_bench.py --calls 1000 --no-frontend | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single', d['extract']['p50_us'], d['extract']['kernel_stage_us'], d['extract']['host_phase_us'])"
done

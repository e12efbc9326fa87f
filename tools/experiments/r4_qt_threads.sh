#!/bin/bash
# K3 (one launch) stage time per step against the threads per workgroup: option quadtree_threads 256 / 512 / 1024 / 0 = the per-launch rule
export TMPDIR=/tmp
for t in 256 512; do PGORB_QT_THREADS=$t timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "quadtree or deep or roots or stages or randomised or errors" 2>&1 | tail -1; done
echo "# K3 ms per step (quadtree_split 0): width height features batch | 256 512 1024 threads | automatic"
run() { PGORB_QT_SPLIT=0 PGORB_QT_THREADS=$5 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap-leg --no-single-frame-leg --no-upload-leg --no-verify --width $1 --height $2 --features $3 --batch $4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['stage_ms_per_step']['quadtree'])"; }
for cfg in "1920 1080 2000 1" "1920 1080 2000 16" "1920 1080 2000 32" "1920 1080 2000 64" "1920 1080 2000 128" "1920 1080 4000 128" "3840 2160 4000 1" "3840 2160 4000 32" "3840 2160 4000 64" "640 480 1000 1" "640 480 1000 64" "640 480 1000 512" "1280 720 1500 128"; do
  set -- $cfg
  echo "$1 $2 $3 $4 | $(run $1 $2 $3 $4 256) $(run $1 $2 $3 $4 512) $(run $1 $2 $3 $4 1024) | $(run $1 $2 $3 $4 0)"
done

#!/bin/bash
# K3 stage time per step, fused (quadtree_split 0) against two launches (1), over frame sizes and batch sizes -> the automatic rule
export TMPDIR=/tmp
echo "# K3 ms per step: width height features batch fused split"
run() { PGORB_QT_SPLIT=$5 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap-leg --no-single-frame-leg --no-upload-leg --no-verify --width $1 --height $2 --features $3 --batch $4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['stage_ms_per_step']['quadtree'])"; }
for cfg in "1920 1080 2000 1" "1920 1080 2000 4" "1920 1080 2000 16" "1920 1080 2000 32" "1920 1080 2000 48" "1920 1080 2000 64" "1920 1080 2000 96" "1920 1080 2000 128" "1920 1080 4000 128" \
           "3840 2160 4000 1" "3840 2160 4000 8" "3840 2160 4000 32" "3840 2160 4000 64" "640 480 1000 1" "640 480 1000 64" "640 480 1000 256" "640 480 1000 512" "1280 720 1500 32" "1280 720 1500 128" "1280 720 1500 256"; do
  set -- $cfg
  echo "$1 $2 $3 $4 $(run $1 $2 $3 $4 0) $(run $1 $2 $3 $4 1)"
done

#!/bin/bash
# K7 at 3840x2160 / 4000 keypoints, batch 32 (31 pairs of ~4 006): the three ways the train descriptors reach the matrix cores + the default choice
export TMPDIR=/tmp
for m in 0 1 2 default; do
  echo "PGORB_MATCH_MODE=$m"
  for i in 1 2 3; do
    if [ $m = default ]; then unset PGORB_MATCH_MODE; else export PGORB_MATCH_MODE=$m; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 0 --width 3840 --height 2160 --features 4000 --batch 32 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   fps %.0f verified %s match %.4f' % (d['value'], d['verified'], d['stage_ms_per_step']['match']))"; done
done

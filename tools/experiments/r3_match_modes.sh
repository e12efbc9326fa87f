#!/bin/bash
# K7: train descriptors expanded once per pair into a slab (mode 0) or taken raw and expanded in LDS by 4-wave (1) / 16-wave (2) / 8-wave (3) workgroups
export TMPDIR=/tmp
for m in ${MODES:-0 1 2 3}; do
  echo "PGORB_MATCH_MODE=$m"
  for i in 1 2 3; do PGORB_MATCH_MODE=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   fps %.0f verified %s match %.4f describe %.4f' % (d['value'], d['verified'], d['stage_ms_per_step']['match'], d['stage_ms_per_step']['describe']))"; done
done

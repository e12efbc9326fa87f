#!/bin/bash
# K2 cells-per-wave sweep (PGORB_FAST_CPW), bench stage time of K2
export TMPDIR=/tmp
for c in 1 2 3 4 6 8 16; do
  PGORB_FAST_CPW=$c python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cpw $c  fast %.4f ms  fps %.0f  verified %s' % (d['stage_ms_per_step']['fast'], d['value'], d['verified']))"
done

#!/bin/bash
# build fast.hip with extra flags ($1) and print K2's stage time (bench, 20 steps); restores nothing (the box is scratch)
export TMPDIR=/tmp
for v in "$@"; do
  touch pilotguru_amd/csrc/fast.hip
  make -C pilotguru_amd/csrc -j8 EXTRA="$v" > /dev/null 2>&1
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v]  fast %.4f ms  fps %.0f  verified %s' % (d['stage_ms_per_step']['fast'], d['value'], d['verified']))"
done

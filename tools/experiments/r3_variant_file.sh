#!/bin/bash
# build with different EXTRA flags for ONE file and print the stage times: r3_variant_file.sh <file.hip> "<flags A>" "<flags B>" ...
export TMPDIR=/tmp
F=$1; shift
for round in 1 2; do
for v in "$@"; do
  touch pilotguru_amd/csrc/$F
  make -C pilotguru_amd/csrc -j8 EXTRA="$v" > /dev/null 2>&1
  for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 0 --no-verify 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v] %s' % {k: round(x, 4) for k, x in d['stage_ms_per_step'].items()})"; done
done; done

#!/bin/bash
# A/B of two versions of one source file on the SAME box: r3_ab.sh <path in tree> <alternative file> [reps]
# (box-to-box spread of K2's time is ~3 %, larger than most single changes)
export TMPDIR=/tmp
F=$1; ALT=$2; REPS=${3:-3}
cp $F /tmp/ab_new
run() { for i in $(seq $REPS); do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 0 --no-verify 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   %s' % {k: round(v, 4) for k, v in d['stage_ms_per_step'].items()})"; done; }
for round in 1 2; do
  cp /tmp/ab_new $F; touch $F; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1; echo "NEW ($F as in the tree):"; run
  cp $ALT $F; touch $F; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1; echo "ALT ($ALT):"; run
done
cp /tmp/ab_new $F

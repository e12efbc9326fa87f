#!/bin/bash
# K2 pair form (PGORB_FAST_PAIRS=1) against the cell form: parity, then K2's stage time
export TMPDIR=/tmp
PGORB_FAST_PAIRS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_shapes.py -m gpu -x -q -k "not sincos and not multi_gpu" 2>&1 | tail -4
for v in 0 1 0 1; do
  PGORB_FAST_PAIRS=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('pairs=$v  fast %.4f ms  fps %.0f  verified %s' % (d['stage_ms_per_step']['fast'], d['value'], d['verified']))"
done

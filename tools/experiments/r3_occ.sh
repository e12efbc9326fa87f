#!/bin/bash
# K2 occupancy sensitivity of the CURRENT kernel: extra LDS per wave (PGORB_FAST_EXTRA_LDS) -> fewer waves per SIMD
export TMPDIR=/tmp
for e in 0 400 1100 1900 3000 5000; do
  PGORB_FAST_EXTRA_LDS=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 0 --no-verify 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('extra LDS $e B (%.0f waves/CU)  fast %.4f ms' % (163840 // (4912 + $e), d['stage_ms_per_step']['fast']))"
done

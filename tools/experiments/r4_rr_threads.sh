#!/bin/bash
# SearchByProjection's rounds kernel with 512 / 256 threads per pair (developer build -DRR_T=...) against the shipped 1024
export TMPDIR=/tmp
for t in 512 256 1024; do
  touch pilotguru_amd/csrc/frame.hip
  make -C pilotguru_amd/csrc -j8 EXTRA="-DRR_T=$t" 2>&1 | grep -E " error"
  echo "== RR_T=$t"
  timeout 300 python -m pytest tests/test_frame_matcher.py -m gpu -x -q -k "projection" 2>&1 | tail -1
  for nf in 2000 4000; do timeout 600 python tools/next_tier_bench.py --batch 128 --features $nf 2>/dev/null | grep -E "SearchByProjection" | cut -c1-140; done
done
touch pilotguru_amd/csrc/frame.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1

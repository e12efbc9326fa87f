#!/bin/bash
# FeatureVectors of a batch: the sorting kernel (default) against the counting kernel (PGORB_FV_COUNTING=1)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_matcher.py tests/test_bench_shapes.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
for nf in 2000 4000; do
  echo "features $nf sorted:   $(timeout 600 python tools/next_tier_bench.py --batch 128 --features $nf 2>/dev/null | grep -E 'FeatureVector' | cut -c1-130)"
  echo "features $nf counting: $(PGORB_FV_COUNTING=1 timeout 600 python tools/next_tier_bench.py --batch 128 --features $nf 2>/dev/null | grep -E 'FeatureVector' | cut -c1-130)"
done

#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_matcher.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -1
for nf in 2000 4000; do timeout 600 python tools/next_tier_bench.py --batch 128 --features $nf 2>/dev/null | grep -E 'SearchByBoW|FeatureVector' | cut -c1-140; done

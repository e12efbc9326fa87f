#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_matcher.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python tools/next_tier_bench.py --batch 128 --features 2000 2>/dev/null | grep -E "SearchForInitialization"
timeout 600 python tools/next_tier_bench.py --batch 128 --features 4000 2>/dev/null | grep -E "SearchForInitialization"

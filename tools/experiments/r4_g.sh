#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
timeout 600 python -m pytest tests/test_frame_matcher.py tests/test_gpu_fuzz.py tests/test_bench_shapes.py -m gpu -x -q -k "not parity[ and not levels and not ingest and not best2" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 600 python tools/next_tier_bench.py --batch 128 --features 4000 --out $O/next_tier_4000.txt > /dev/null 2>&1; cat $O/next_tier_4000.txt
timeout 600 python tools/next_tier_bench.py --batch 128 --features 2000 --out $O/next_tier_2000.txt > /dev/null 2>&1; cat $O/next_tier_2000.txt

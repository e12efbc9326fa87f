#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_frame_matcher.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not parity and not levels and not ingest and not best2" 2>&1 | tail -15 > gpurun_out/r4b/pytest.txt
cat gpurun_out/r4b/pytest.txt
python tools/single_frame_bench.py --calls 1000 --out gpurun_out/r4b/single_frame_before.json

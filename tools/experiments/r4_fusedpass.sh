#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
bash tools/experiments/r4_split_grid.sh 2>/dev/null
bash tools/experiments/r4_qtp_timing.sh 2>&1 | grep -v amdgpu | head -30

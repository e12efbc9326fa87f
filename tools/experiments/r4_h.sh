#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt

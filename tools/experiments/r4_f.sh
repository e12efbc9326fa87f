#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
timeout 120 python tools/experiments/r4_k3_check.py 2>&1 | grep -v amdgpu.ids | tail -3
for kpw in 0 4 16; do PGORB_DESC_KPW=$kpw timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg --sustain-seconds 0 --no-verify 2>&1 | tail -1 > $O/bench_line.json
python -c "
import json; d=json.load(open('$O/bench_line.json')); print($kpw, d['value'], d['stage_ms_per_step'])"; done
bash tools/experiments/r4_desc_timing.sh

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
timeout 120 python tools/experiments/r4_k3_check.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not sincos and not ingest and not best2 and not matchers" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for kpw in 0 4; do echo "kpw $kpw"; PGORB_DESC_KPW=$kpw timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg 2>&1 | tail -1 > $O/bench_line.json
python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d.get('sustained_fps'), d['stage_ms_per_step'], d['verified'])"; done
timeout 300 python bench.py --width 3840 --height 2160 --features 4000 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg 2>&1 | tail -1 > $O/bench_line_4k.json
python -c "
import json; d=json.load(open('$O/bench_line_4k.json')); print('4k', d['value'], d.get('sustained_fps'), d['stage_ms_per_step'], d['verified'])"

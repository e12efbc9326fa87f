#!/bin/bash
# after the K3 split + rule: the whole GPU suite, bench lines at the three shapes, the single frame
export TMPDIR=/tmp
mkdir -p gpurun_out/r4i
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4i/pytest.txt 2>&1; tail -4 gpurun_out/r4i/pytest.txt
timeout 300 python bench.py --no-cpu-baseline --no-single-frame-leg > gpurun_out/r4i/bench_line.json 2>/dev/null; python -c "import json; d=json.loads(open('gpurun_out/r4i/bench_line.json').read().strip().splitlines()[-1]); print('1080p', round(d['value']), d['verified'], d['stage_ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-single-frame-leg --width 3840 --height 2160 --features 4000 --batch 32 > gpurun_out/r4i/bench_line_4k.json 2>/dev/null; python -c "import json; d=json.loads(open('gpurun_out/r4i/bench_line_4k.json').read().strip().splitlines()[-1]); print('2160p', round(d['value']), d['verified'], d['stage_ms_per_step'])"
timeout 300 python tools/single_frame_bench.py --calls 2000 --out gpurun_out/r4i/single_frame.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single', d['extract'], d.get('c_abi_extract_grid_search_for_initialization'), d.get('c_abi_extract_grid_bow_transform'))"

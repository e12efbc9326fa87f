#!/bin/bash
# every step under its own timeout: a kernel that hangs must not hold the box until gpurun's limit
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_shapes.py tests/test_gpu_fuzz.py tests/test_cli.py -m gpu -x -q -k "not sincos" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 120 python tools/single_frame_bench.py --calls 1000 --out $O/single_frame_graph.json > /dev/null 2>&1
PGORB_EXTRACT_NO_GRAPH=1 timeout 120 python tools/single_frame_bench.py --calls 1000 --no-frontend --out $O/single_frame_nograph.json > /dev/null 2>&1
python - <<'PY'
import json
for f in ("graph", "nograph"):
    try: print(f, json.load(open("gpurun_out/r4k/single_frame_%s.json" % f)))
    except Exception as e: print(f, "missing", e)
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg 2>&1 | tail -1 > $O/bench_line.json
python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d.get('sustained_fps'), d['stage_ms_per_step'], d['verified'])"
timeout 300 python bench.py --width 3840 --height 2160 --features 4000 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg 2>&1 | tail -1 > $O/bench_line_4k.json
python -c "
import json; d=json.load(open('$O/bench_line_4k.json')); print('4k', d['value'], d.get('sustained_fps'), d['stage_ms_per_step'], d['verified'])"

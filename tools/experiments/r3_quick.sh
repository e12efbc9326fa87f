#!/bin/bash
# round 3 iteration loop: K2-facing parity tests, then the bench's stage times (one line)
export TMPDIR=/tmp
mkdir -p gpurun_out/r3q
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_shapes.py -m gpu -x -q -k "not sincos and not multi_gpu" 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --sustain-seconds 2 2>&1 | tail -1 > gpurun_out/r3q/line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3q/line.json"))
print("fps %.0f sustained %.0f verified %s stages %s" % (d["value"], d.get("sustained_fps", 0), d["verified"], {k: round(v, 4) for k, v in d["stage_ms_per_step"].items()}))
PY

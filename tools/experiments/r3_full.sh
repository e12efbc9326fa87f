#!/bin/bash
# full GPU suite, then the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py 2>gpurun_out/r3f/bench.err | tail -1 > gpurun_out/r3f/line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3f/line.json"))
print("fps %.0f sustained %.0f verified %s stages %s" % (d["value"], d.get("sustained_fps", 0), d["verified"], {k: round(v, 4) for k, v in d["stage_ms_per_step"].items()}))
u = d["frames_uploaded"]
print("uploaded %.0f (%.2f of link %.1f GB/s)  rgb24 %.0f (%.2f of link/3)  frontend %.0f" % (u["value"], u["fraction_of_link"], u["h2d_GBps_link"], u["rgb24"]["value"], u["rgb24"]["fraction_of_link"], u["with_front_end_stage"]["value"]))
print("two batches", d["two_batches_in_flight"]["fps"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["one_thread"]["value"])
PY

#!/bin/bash
# A >= 5 s sustained run of the bench step with the GPU's clocks / power / utilisation sampled beside it
# (rocm-smi every 0.5 s).  usage: tools/sustained_run.sh <tag>   -> gpurun_out/<tag>_sustained.txt
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_sustained.txt
mkdir -p gpurun_out
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" ; sleep 0.5; done ) > gpurun_out/${TAG}_smi.log 2>&1 &
SMI=$!
python bench.py --steps 20 --warmup 5 --sustain-seconds ${SUSTAIN:-6} --no-cpu-baseline --no-upload-leg --no-overlap-leg > gpurun_out/${TAG}_sustained_line.json 2> gpurun_out/${TAG}_sustained.err
kill $SMI 2>/dev/null
{
  echo "# python bench.py --steps 20 --warmup 5 --sustain-seconds ${SUSTAIN:-6} --no-cpu-baseline --no-upload-leg --no-overlap-leg ; rocm-smi sampled every 0.5 s beside it"
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_sustained_line.json").read().strip().splitlines()[-1])
print("timed 20 steps: %.0f frames/s (%.3f ms/step); sustained %.0f frames/s over %.2f s (%d steps); verified %s" % (
    d["value"], d["ms_per_step"], d["sustained_fps"], d["sustained"]["seconds"], d["sustained"]["steps"], d["verified"]))
print("stage ms per step:", {k: round(v, 4) for k, v in d["stage_ms_per_step"].items()})
PY
  echo "# rocm-smi samples (sclk / mclk / power / GPU use), idle -> load -> idle:"
  grep -E "sclk|mclk|Power|GPU use" gpurun_out/${TAG}_smi.log | sed 's/^=*//' | awk '{$1=$1};1' | paste - - - - | awk "NR % ${EVERY:-1} == 0" | head -60
} > $OUT
cat $OUT

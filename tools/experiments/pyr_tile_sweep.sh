#!/bin/bash
# Tile-size sweep of the LDS-staged pyramid kernel K1 (BASELINE.json configs[2]): 256 x {16, 32, 64} destination
# tiles at 1080p / 2000 kp (batch 128) and 3840x2160 / 4000 kp (batch 32); parity first, then the stage time.
#   gpurun -- tools/experiments/pyr_tile_sweep.sh
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pyr_sweep; mkdir -p $OUT
for R in 16 32 64; do
  export PGORB_PYR_TILE_ROWS=$R
  python -m pytest tests/test_gpu_parity.py -x -q -k "pyramid or stage or extract" 2>&1 | tail -1 > $OUT/parity_$R.txt
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/hd_$R.json
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --width 3840 --height 2160 --features 4000 --batch 32 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/4k_$R.json
done
unset PGORB_PYR_TILE_ROWS
PGORB_PYR_NO_LDS=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/hd_nolds.json
PGORB_PYR_NO_LDS=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --width 3840 --height 2160 --features 4000 --batch 32 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/4k_nolds.json
python - <<'PY'
import json
print("# K1 tile sweep: destination tile 256 x R; pyramid stage ms per step (HIP events), whole-path frames/s")
print("%-10s %-28s %-12s %-12s %-12s %-12s" % ("tile", "parity", "1080p K1 ms", "1080p fps", "4K K1 ms", "4K fps"))
for R in ("16", "32", "64", "nolds"):
    hd = json.loads(open("gpurun_out/pyr_sweep/hd_%s.json" % R).read()); k4 = json.loads(open("gpurun_out/pyr_sweep/4k_%s.json" % R).read())
    par = open("gpurun_out/pyr_sweep/parity_%s.txt" % R).read().strip() if R != "nolds" else "(register windows, no LDS)"
    print("%-10s %-28s %-12.3f %-12.0f %-12.3f %-12.0f" % ("256x" + R if R != "nolds" else "no LDS", par, hd["stage_ms_per_step"]["pyramid"], hd["value"], k4["stage_ms_per_step"]["pyramid"], k4["value"]))
PY

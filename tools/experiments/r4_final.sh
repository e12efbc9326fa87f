#!/bin/bash
# Round-4 closing measurements (GPU box): changed host paths re-tested, K1's launches one row per pyramid level, the
# BASELINE table lines, the sustained run, the CLI rate, the single-frame leg.  Every step under its own timeout.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_frame_matcher.py tests/test_bow.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4_final_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r4_final_tests.log
tail -3 gpurun_out/r4_final_tests.log
# K1 per level
for cfg in "1080p" "2160p --width 3840 --height 2160 --features 4000 --batch 32"; do
  set -- $cfg; tag=$1; shift
  rm -rf /tmp/k1prof; timeout 600 rocprofv3 --kernel-trace -d /tmp/k1prof -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-leg --no-single-frame-leg --no-upload-leg "$@" > /tmp/k1prof.log 2>&1
  db=$(find /tmp/k1prof -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace -- python bench.py --steps 20 --warmup 5 $* ; K1's launches, one row per launch shape (= pyramid level); us"
    python tools/rocpd_summary.py bygrid "$db" k_pyr; python tools/rocpd_summary.py bygrid "$db" k_copy_level0; } > gpurun_out/r04_k1_levels_$tag.txt 2>&1
  cat gpurun_out/r04_k1_levels_$tag.txt
done
rm -rf /tmp/k1prof
timeout 1500 bash tools/baseline_table.sh r04_end
SUSTAIN=45 EVERY=6 timeout 400 bash tools/sustained_run.sh r04_end
timeout 600 bash tools/experiments/cli_rate.sh 2048 > gpurun_out/r04_cli_rate.txt 2>&1; tail -30 gpurun_out/r04_cli_rate.txt
timeout 600 python tools/single_frame_bench.py --calls 2000 --out gpurun_out/r04_single_frame.json | tail -60
timeout 900 python tools/next_tier_bench.py --batch 128 --features 2000 > gpurun_out/r04_next_tier.txt 2> gpurun_out/r04_next_tier.err; tail -25 gpurun_out/r04_next_tier.txt
timeout 900 python tools/next_tier_bench.py --batch 128 --features 4000 > gpurun_out/r04_next_tier_init4000.txt 2> gpurun_out/r04_next_tier4000.err; tail -25 gpurun_out/r04_next_tier_init4000.txt
find gpurun_out -name '*.db' -delete

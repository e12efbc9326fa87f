#!/bin/bash
# closing profiles of round 4 (final build): rocprofv3 kernel stats + PMC groups at 1080p (r04_b) and 2160p (r04_b4k), then r4_final.sh
export TMPDIR=/tmp
timeout 1500 bash tools/profile_round.sh r04_b > gpurun_out/r04_b.log 2>&1; tail -3 gpurun_out/r04_b.log
timeout 1500 bash tools/profile_round.sh r04_b4k --width 3840 --height 2160 --features 4000 --batch 32 > gpurun_out/r04_b4k.log 2>&1; tail -3 gpurun_out/r04_b4k.log
find gpurun_out -name '*.db' -delete
bash tools/experiments/r4_final.sh

#!/bin/bash
export TMPDIR=/tmp
timeout 1500 bash tools/profile_round.sh r04_a > gpurun_out/r04_a.log 2>&1
timeout 1500 bash tools/profile_round.sh r04_a4k --width 3840 --height 2160 --features 4000 --batch 32 > gpurun_out/r04_a4k.log 2>&1
du -sh gpurun_out; ls gpurun_out/r04_a gpurun_out/r04_a4k
timeout 900 bash tools/experiments/k2_tile_sweep.sh 3840,2160,4000,32 > /dev/null 2>&1
timeout 900 bash tools/experiments/k2_tile_sweep.sh 1920,1080,2000,128 > /dev/null 2>&1
rm -rf gpurun_out/k2sw_*
cat gpurun_out/k2_sweep_3840x2160.txt

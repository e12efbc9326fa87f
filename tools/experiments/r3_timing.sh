#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3t
touch pilotguru_amd/csrc/fast.hip
make -C pilotguru_amd/csrc -j8 EXTRA=-DPGORB_FAST_TIMING > gpurun_out/r3t/build.log 2>&1
python tools/experiments/fast_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3t/fast_timing.txt

#!/bin/bash
# round 3, probe 1: issue costs of the byte-parallel candidates, and where a K2 wave's life goes
export TMPDIR=/tmp
mkdir -p gpurun_out/r3p1
make -C tools/ubench valu_rate4 > /dev/null 2>&1 && tools/ubench/valu_rate4 > gpurun_out/r3p1/valu_rate4.txt 2>&1
touch pilotguru_amd/csrc/fast.hip
make -C pilotguru_amd/csrc -j8 EXTRA=-DPGORB_FAST_TIMING > gpurun_out/r3p1/build.log 2>&1
python tools/experiments/fast_timing.py > gpurun_out/r3p1/fast_timing.txt 2>&1
cat gpurun_out/r3p1/valu_rate4.txt gpurun_out/r3p1/fast_timing.txt

#!/bin/bash
# K3 phase times of the (frame 0, level l) workgroup for l = 0, 3, 7 at batch 1 and 128 (developer build -DPGORB_QT_TIMING -DQT_TIMING_LEVEL=l)
export TMPDIR=/tmp
for l in 0 3 7; do
  touch pilotguru_amd/csrc/quadtree.hip
  make -C pilotguru_amd/csrc -j8 EXTRA="-DPGORB_QT_TIMING -DQT_TIMING_LEVEL=$l" 2>&1 | grep -E "error" 
  echo "== level $l"; python tools/experiments/qt_timing.py 2>&1 | grep -v "amdgpu.ids"
done
touch pilotguru_amd/csrc/quadtree.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1

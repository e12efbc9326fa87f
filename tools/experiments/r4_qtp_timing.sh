#!/bin/bash
export TMPDIR=/tmp
touch pilotguru_amd/csrc/quadtree.hip
make -C pilotguru_amd/csrc -j8 EXTRA="-DPGORB_QT_TIMING -DQT_TIMING_LEVEL=0" 2>&1 | grep -E "error"
timeout 300 python tools/experiments/qt_timing.py 2>&1 | grep -v "amdgpu.ids"
touch pilotguru_amd/csrc/quadtree.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1

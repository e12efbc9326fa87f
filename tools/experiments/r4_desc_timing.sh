#!/bin/bash
export TMPDIR=/tmp
touch pilotguru_amd/csrc/describe.hip
make -C pilotguru_amd/csrc -j8 EXTRA=-DPGORB_DESC_TIMING 2>&1 | grep error
timeout 300 python tools/experiments/desc_timing.py 2>&1 | grep -v amdgpu.ids
touch pilotguru_amd/csrc/describe.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1

#!/bin/bash
export TMPDIR=/tmp
build() { touch pilotguru_amd/csrc/quadtree.hip; make -C pilotguru_amd/csrc -j8 EXTRA="$1" 2>&1 | grep -E "error" ; }
for v in ""; do
  echo "=== variant [$v]"; build "$v"
  PGORB_EXTRACT_NO_GRAPH=1 timeout 120 python tools/experiments/r4_k3_check.py 2>&1 | grep -v amdgpu.ids | tail -5
  echo "--- with graph"; timeout 120 python tools/experiments/r4_k3_check.py 2>&1 | grep -v amdgpu.ids | tail -5
done
build ""

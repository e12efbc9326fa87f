#!/bin/bash
# BASELINE.json configs[2]'s tile sweep for K2: every tile shape through the bench workload, K2 time from HIP events
# (k2_tile_sweep.py) and HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one run each).
# usage: tools/experiments/k2_tile_sweep.sh [W,H,NF,B]    -> gpurun_out/k2_sweep_<W>x<H>.txt
export TMPDIR=/tmp
# the block form is a developer build since round 3
touch pilotguru_amd/csrc/fast.hip pilotguru_amd/csrc/api.hip; make -C pilotguru_amd/csrc -j8 EXTRA=-DPGORB_FAST_BLOCKS > /dev/null 2>&1
CFG=${1:-1920,1080,2000,128}
W=$(echo $CFG | cut -d, -f1); H=$(echo $CFG | cut -d, -f2)
OUT=gpurun_out/k2_sweep_${W}x${H}.txt
mkdir -p gpurun_out
{
echo "# K2 tile sweep, $CFG (width,height,features,batch): HIP-event time per launch, algorithmic GB/s, then PMC traffic"
PGORB_SWEEP_CFG=$CFG python tools/experiments/k2_tile_sweep.py
for sh in "0,0" "1,1" "2,2" "4,2" "4,4"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/k2sw_${sh/,/_}_$ctr; rm -rf $d
    PGORB_SWEEP_CFG=$CFG PGORB_SWEEP_SHAPES="$sh" rocprofv3 --kernel-trace --pmc $ctr -d $d -- python tools/experiments/k2_tile_sweep.py > /dev/null 2>&1
    db=$(find $d -name '*.db' | head -1)
    echo "shape $sh $ctr: $(python tools/rocpd_summary.py pmc $db | grep -E 'k_fast' | awk '{print $1, $3}')"
  done
done
} | tee $OUT

#!/bin/bash
# BASELINE.json configs[2]'s tile sweep for K2 (the shipped kernel's LDS window pitch x waves per workgroup): K2 time from HIP events
# (k2_tile_sweep.py) and HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one run each).
# usage: tools/experiments/k2_tile_sweep.sh [W,H,NF,B]    -> gpurun_out/k2_sweep_<W>x<H>.txt
export TMPDIR=/tmp
CFG=${1:-1920,1080,2000,128}
W=$(echo $CFG | cut -d, -f1); H=$(echo $CFG | cut -d, -f2)
OUT=gpurun_out/k2_sweep_${W}x${H}.txt
mkdir -p gpurun_out
{
echo "# K2 tile sweep of the shipped kernel, $CFG (width,height,features,batch): HIP-event time per launch, algorithmic GB/s; then PMC traffic"
echo "# (HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB, MI355X_MICROARCH.md HBM section; algorithmic = sum of the level planes x batch)"
PGORB_SWEEP_CFG=$CFG timeout 600 python tools/experiments/k2_tile_sweep.py 2>&1 | grep -v amdgpu.ids
for sh in "0,1" "64,1" "96,1" "128,1" "0,4"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/k2sw_${sh/,/_}_$ctr; rm -rf $d
    PGORB_SWEEP_CFG=$CFG PGORB_SWEEP_SHAPES="$sh" timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $d -- python tools/experiments/k2_tile_sweep.py > /dev/null 2>&1
    db=$(find $d -name '*.db' | head -1)
    echo "shape $sh $ctr (KB per launch): $(python tools/rocpd_summary.py pmc $db | grep -E 'k_fast' | awk '{print $1, $3}')"
  done
done
} | tee $OUT

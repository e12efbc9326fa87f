#!/bin/bash
# PMC passes over one AccelerometerCalibrator evaluation on the GPU (k_calibrate_windows, one wave):
#   gpurun -- tools/profile_calib.sh <tag>
set -e
TAG=${1:-calib}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
CMD="python tools/experiments/calib_eval_latency.py 100"
{
  echo "# rocprofv3 --kernel-trace --pmc <group> -- $CMD ; per-dispatch means over the k_calibrate_windows launches (1, 1, 256, 1024, 4096 waves)"
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_IFETCH"; do
    n=$(echo "$grp" | tr ' ' '_' | cut -c1-24)
    rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc_$n" -- $CMD > "$OUT/pmc_$n.log" 2>&1 || true
    db=$(find "$OUT/pmc_$n" -name '*.db' | head -1)
    echo; echo "## --pmc $grp"
    [ -n "$db" ] && python tools/rocpd_summary.py pmc "$db" | grep -v "rocclr" || echo "(no output)"
  done
} > "$OUT/${TAG}_pmc.txt"
cat "$OUT/${TAG}_pmc.txt"

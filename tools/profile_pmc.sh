#!/bin/bash
# PMC passes for the bench workload (run on the GPU box through gpurun).  Counters go in their
# own runs with --kernel-trace only (no sys/hip/hsa tracing together with --pmc).
# usage: tools/profile_pmc.sh <outdir-under-gpurun_out> [bench args...]
set -e
OUT=${1:-gpurun_out/pmc}; shift || true
export TMPDIR=/tmp
mkdir -p "$OUT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/sq1" -- python bench.py $ARGS > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM -d "$OUT/sq2" -- python bench.py $ARGS > "$OUT/sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS -d "$OUT/sq3" -- python bench.py $ARGS > "$OUT/sq3.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -- python bench.py $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -- python bench.py $ARGS > "$OUT/write.log" 2>&1
for d in sq1 sq2 sq3 fetch write; do
  db=$(find "$OUT/$d" -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py pmc "$db" > "$OUT/$d.txt" 2>&1 || echo "no db for $d" > "$OUT/$d.txt"
done
cat "$OUT"/sq1.txt "$OUT"/sq2.txt "$OUT"/sq3.txt "$OUT"/fetch.txt "$OUT"/write.txt

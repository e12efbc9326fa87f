#!/bin/bash
# round 4, first GPU call: the whole GPU suite on the refactored options / fuzz slices, a baseline bench line, and the
# kernel trace of the next-tier matchers at 2000 and 4000 features (which kernel owns the 4000-feature cliff?)
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4a/pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r4a/bench_line.json
for nf in 2000 4000; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r4a/nt$nf -- python tools/next_tier_bench.py --batch 128 --features $nf --out gpurun_out/r4a/next_tier_$nf.txt > gpurun_out/r4a/nt$nf.log 2>&1
  db=$(find gpurun_out/r4a/nt$nf -name '*.db' | head -1)
  python tools/rocpd_summary.py stats "$db" > gpurun_out/r4a/nt${nf}_kernel_stats.txt 2>&1
done
cat gpurun_out/r4a/pytest.txt; cat gpurun_out/r4a/bench_line.json; cat gpurun_out/r4a/nt4000_kernel_stats.txt

#!/bin/bash
# round 6 soak (final build; the fuzzers alternate the fused level launch and K1 + K2): every fuzzer of tests/fuzzers.py with fresh seeds -> gpurun_out/r06_fuzz_soak.txt
export TMPDIR=/tmp
O=gpurun_out/r06_fuzz_soak.txt
{
echo "# tools/experiments/r6_soak.sh on one MI355X (round-6 build: fused_levels alternated by the fuzzers): random configurations, HIP path against the oracle, bit for bit"
for spec in "parity 5000 601" "levels 200 602" "batch_parity 1200 603" "matchers 1500 604" "ingest 500 605" "best2 3000 606"; do
  set -- $spec
  echo "## fuzz_$1 $2 cases, seed $3"
  timeout 1500 python tools/experiments/fuzz_$1.py $2 $3 2>&1 | grep -v "amdgpu.ids\|^skip" | tail -4
done
echo "## matchers at the initialisation workload (1920x1080, 4000 features: dense windows), 40 cases, seed 607"
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import fuzzers
print(fuzzers.fuzz_matchers(cases=40, seed=607, nf_range=(4000, 4001), size_range=((1920, 1921), (1080, 1081)))[1])
PY
} > $O 2>&1
cat $O
{
echo "## large frames (1100..2600 x 700..1500, 500..5000 features): the levels with more cells than threads, K3 forms and thread counts rotated; fuzz_parity 160 cases seed 608, fuzz_batch_parity 40 cases seed 609"
timeout 1500 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import fuzzers
print(fuzzers.fuzz_parity(cases=160, seed=608, size_range=((1100, 2600), (700, 1500)), nf_range=(500, 5000))[1])
print(fuzzers.fuzz_batch_parity(cases=40, seed=609, size_range=((1100, 2000), (700, 1200)), nf_range=(500, 4000))[1])
PY
} >> $O 2>&1
tail -8 $O

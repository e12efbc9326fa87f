#!/bin/bash
# round 6, first GPU call: the fused resize + detect launch against the oracle, then K1 + K2 vs fused at bench size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or stages_and_output" 2>&1 | tail -15 > gpurun_out/r6_first_tests.txt
for f in 0 1; do
  timeout 300 python bench.py --steps 20 --warmup 5 --fused-levels $f --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --sustain-seconds 1 > gpurun_out/r6_first_bench_f$f.json 2> gpurun_out/r6_first_bench_f$f.err
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6_first_prof -o fused -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --no-verify --sustain-seconds 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r6_first_prof -name "*kernel_stats*" | head; cat gpurun_out/r6_first_tests.txt; for f in 0 1; do python -c "
import json,sys
l=[x for x in open('gpurun_out/r6_first_bench_f$f.json') if x.startswith('{')]
if not l: print('no line', open('gpurun_out/r6_first_bench_f$f.err').read()[-1500:]); sys.exit()
o=json.loads(l[-1]); print('fused=$f', o['value'], o['ms_per_step'], o['stage_ms_per_step'], o['verified'], o['roofline']['kernel'], o['roofline']['frac'])
"; done

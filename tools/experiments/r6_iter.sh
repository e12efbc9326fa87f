#!/bin/bash
# round 6 iteration loop: fused tests (short), phase timing of the timing build, bench A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused" 2>&1 | tail -5
[ -f pilotguru_amd/libpgorb_timing.so ] && PGORB_LIBRARY=$PWD/pilotguru_amd/libpgorb_timing.so timeout 300 python tools/experiments/r6_fuse_timing.py 2>&1 | tail -8
for f in ${FUSED_SET:-1}; do
  timeout 300 python bench.py --steps 20 --warmup 5 --fused-levels $f --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --sustain-seconds 1 > gpurun_out/r6_iter_bench_f$f.json 2> gpurun_out/r6_iter_bench_f$f.err
  python -c "
import json,sys
l=[x for x in open('gpurun_out/r6_iter_bench_f$f.json') if x.startswith('{')]
if not l: print('no line', open('gpurun_out/r6_iter_bench_f$f.err').read()[-1500:]); sys.exit()
o=json.loads(l[-1]); print('fused=$f', round(o['value']), round(o['ms_per_step'],4), {k: round(v,4) for k,v in o['stage_ms_per_step'].items()}, o['verified'])
"
done

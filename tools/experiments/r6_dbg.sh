#!/bin/bash
# fused launch: what its parts cost (PGORB_FUSE_DBG: 1 = no resize arithmetic (K1 runs in front), 2 = no detection), + per-level kernel times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for d in 0 1 2 3; do
  PGORB_FUSE_DBG=$d timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --no-verify --sustain-seconds 1 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
o=json.loads(l[-1]); print('dbg=$d', round(o['value']), round(o['ms_per_step'],4), {k: round(v,4) for k,v in o['stage_ms_per_step'].items()})
"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6_dbg_prof -o fused -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --no-verify --sustain-seconds 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py bygrid gpurun_out/r6_dbg_prof/fused_results.db k_fast_resize

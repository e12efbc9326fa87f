#!/bin/bash
# stage times per FRAME against the batch size (is a pyramid that fits the 256-MB Infinity Cache cheaper to read back?)
cd "$GRAFT_REPO_ROOT"
for b in 16 32 48 64 128 256; do
  timeout 300 python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --no-upload-leg --no-overlap-leg --no-single-frame-leg --no-traffic-leg --no-verify --sustain-seconds 1 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
o=json.loads(l[-1]); b=$b
print('batch %3d  %7.0f frames/s  us per frame: ' % (b, o['value']) + '  '.join('%s %.2f' % (k, v * 1e3 / b) for k, v in o['stage_ms_per_step'].items()) + '  step %.2f' % (o['ms_per_step'] * 1e3 / b))
"
done

#!/bin/bash
# BASELINE.md's results table: the three single-GPU configurations through bench.py (CPU legs included),
# the driving-like scene, one line each under gpurun_out/<tag>/ (copy the ones to keep into profiles/).
TAG=${1:-r02_table}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --width 640 --height 480 --features 1000 --batch 512 > $OUT/${TAG}_640x480_1000.json 2> $OUT/err_640.log
python bench.py > $OUT/${TAG}_1920x1080_2000.json 2> $OUT/err_1080.log
python bench.py --width 3840 --height 2160 --features 4000 --batch 32 > $OUT/${TAG}_3840x2160_4000.json 2> $OUT/err_4k.log
python bench.py --scene road > $OUT/${TAG}_1920x1080_2000_road.json 2> $OUT/err_road.log
for f in $OUT/*.json; do echo "== $f"; tail -1 $f | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d.get('cpu_baseline', {})
print('fps %.0f  sustained %.0f  uploaded %.0f  frac %.3f  kp %.0f  verified %s  cpu1 %.2f  cpuN %.1f (n=%s)  stages %s' % (
    d['value'], d.get('sustained_fps', 0), d.get('frames_uploaded', {}).get('value', 0), d['roofline']['frac'],
    d['config']['keypoints_per_frame'], d['verified'], c.get('one_thread', {}).get('value', 0), c.get('value', 0), c.get('cores'),
    {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}))"; done

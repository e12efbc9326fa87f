#!/bin/bash
# Throughput of the drop-in CLI (front-end mode) on a 1080p ride held in /dev/shm: frames/s end to end.
set -e
D=/dev/shm/pgcli; rm -rf $D; mkdir -p $D
N=${1:-256}
NR=$(( N < 512 ? N : 512 ))          # the RGB24 clip is three times the bytes: at most 512 frames of it
python - <<PY
import sys; sys.path.insert(0, '.')
from pilotguru_amd.synth import synth_ride
from pilotguru_amd import vocab as V
r = synth_ride(0, 1920, 1080, 32)
import numpy as np
with open('$D/clip.gray', 'wb') as f:
    for k in range($N // 32): r.tofile(f)
rgb = np.ascontiguousarray(np.stack([r, r, r], axis=3))          # RGB24, what the reference's reader decodes to
with open('$D/clip.rgb', 'wb') as f:
    for k in range($NR // 32): rgb.tofile(f)
open('$D/cam.yml', 'w').write("%YAML:1.0\n---\nCamera_width: 1920\nCamera_height: 1080\nCamera_fps: 30.\nORBextractor_nFeatures: 2000\n")
d, w, p = V.synth_vocabulary(10, 4, seed=5)
V.write_vocabulary_text('$D/voc.txt', 10, 4, d, w, p)
PY
for b in 8 32 64 128; do
  t0=$(date +%s.%N)
  pilotguru_amd/host/optical_trajectories --vocabulary_file=$D/voc.txt --camera_settings=$D/cam.yml \
    --in_video=$D/clip.gray --out_dir=$D --novisualize --batch=$b 2>&1 | tail -1
  t1=$(date +%s.%N)
  python -c "print('batch $b: %.2f s wall, %.0f frames/s (incl. process start, vocabulary load, context creation)' % ($t1 - $t0, $N / ($t1 - $t0)))"
done
for b in 64; do
  pilotguru_amd/host/optical_trajectories --vocabulary_file=$D/voc.txt --camera_settings=$D/cam.yml \
    --in_video=$D/clip.rgb --out_dir=$D --novisualize --batch=$b --vertical_flip 2>&1 | tail -1 | sed 's/^/RGB24 + --vertical_flip (on the device), batch 64: /'
  pilotguru_amd/host/optical_trajectories --vocabulary_file=$D/voc.txt --camera_settings=$D/cam.yml \
    --in_video=$D/clip.gray --out_dir=$D --novisualize --batch=$b --vertical_flip --horizontal_flip 2>&1 | tail -1 | sed 's/^/grey + both flips (on the device), batch 64: /'
done
rm -rf $D

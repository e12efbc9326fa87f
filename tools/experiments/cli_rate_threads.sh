#!/bin/bash
# CLI frame-loop rate against --copy_threads and --batch (2048 grey / 1024 RGB24 1080p frames in /dev/shm)
set -e
D=/dev/shm/pgcli2; rm -rf $D; mkdir -p $D
python - <<PY
import sys; sys.path.insert(0, '.')
import numpy as np
from pilotguru_amd.synth import synth_ride
from pilotguru_amd import vocab as V
r = synth_ride(0, 1920, 1080, 32)
with open('$D/clip.gray', 'wb') as f:
    for k in range(64): r.tofile(f)
rgb = np.ascontiguousarray(np.stack([r, r, r], axis=3))
with open('$D/clip.rgb', 'wb') as f:
    for k in range(32): rgb.tofile(f)
open('$D/cam.yml', 'w').write("%YAML:1.0\n---\nCamera_width: 1920\nCamera_height: 1080\nCamera_fps: 30.\nORBextractor_nFeatures: 2000\n")
d, w, p = V.synth_vocabulary(10, 4, seed=5)
V.write_vocabulary_text('$D/voc.txt', 10, 4, d, w, p)
PY
for clip in gray rgb; do for t in 4 8 16 32; do for b in 32 64 128; do
  pilotguru_amd/host/optical_trajectories --vocabulary_file=$D/voc.txt --camera_settings=$D/cam.yml --in_video=$D/clip.$clip --out_dir=$D \
    --novisualize --batch=$b --copy_threads=$t 2>&1 | tail -1 | sed "s/.*frame loop: /$clip threads $t batch $b: /"
done; done; done
rm -rf $D

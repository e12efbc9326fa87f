#!/bin/bash
# Throughput of the drop-in CLI (front-end mode) on a 1080p ride held in /dev/shm: frames/s end to end, and where the start-up goes.
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
open('$D/cam.yml', 'w').write("%YAML:1.0\n---\nCamera_width: 1920\nCamera_height: 1080\nCamera_fps: 30.\nCamera_RGB: 1\nORBextractor_nFeatures: 2000\n")
d, w, p = V.synth_vocabulary(10, 4, seed=5)
V.write_vocabulary_text('$D/voc.txt', 10, 4, d, w, p)
d, w, p = V.synth_vocabulary_fast(10, 6, seed=7)                 # ORBvoc-sized: k = 10, L = 6, 1.1 M nodes, ~145 MB of text
V.write_vocabulary_text_fast('$D/vocbig.txt', 10, 6, d, w, p)
PY
run() { t0=$(date +%s.%N); PGORB_CLI_TIMING=1 pilotguru_amd/host/optical_trajectories "$@" > $D/run.log 2>&1; t1=$(date +%s.%N); tail -2 $D/run.log
        # where the wall clock goes outside main(): exec + dynamic loading before it, the kernel's teardown of the process after _exit
        m=$(grep -o "main entered at epoch [0-9.]*" $D/run.log | grep -o "[0-9.]*$"); e=$(grep -o "epoch [0-9.]*$" $D/run.log | tail -1 | grep -o "[0-9.]*$")
        python -c "print('    -> %.3f s wall (%.3f s before main, %.3f s between the report and the shell seeing the exit), %.0f frames/s including process start' % ($t1 - $t0, $m - $t0, $t1 - $e, $N / ($t1 - $t0)))"; }
for b in 8 32 64 128; do
  echo "grey, batch $b:"; run --vocabulary_file=$D/voc.txt --camera_settings=$D/cam.yml --in_video=$D/clip.gray --out_dir=$D --novisualize --batch=$b
done
echo "grey, batch 64, second run (vocabulary cache present):"; run --vocabulary_file=$D/voc.txt --camera_settings=$D/cam.yml --in_video=$D/clip.gray --out_dir=$D --novisualize --batch=64
echo "grey, batch 64, ORBvoc-sized vocabulary, text parsed (--novocabulary_cache):"; run --vocabulary_file=$D/vocbig.txt --camera_settings=$D/cam.yml --in_video=$D/clip.gray --out_dir=$D --novisualize --batch=64 --novocabulary_cache
echo "grey, batch 64, ORBvoc-sized vocabulary, first run with the cache (parses + writes it):"; run --vocabulary_file=$D/vocbig.txt --camera_settings=$D/cam.yml --in_video=$D/clip.gray --out_dir=$D --novisualize --batch=64
echo "grey, batch 64, ORBvoc-sized vocabulary, from its cache:"; run --vocabulary_file=$D/vocbig.txt --camera_settings=$D/cam.yml --in_video=$D/clip.gray --out_dir=$D --novisualize --batch=64
N=$NR
echo "RGB24 + --vertical_flip (on the device), batch 64, $NR frames:"; run --vocabulary_file=$D/voc.txt --camera_settings=$D/cam.yml --in_video=$D/clip.rgb --out_dir=$D --novisualize --batch=64 --vertical_flip
rm -rf $D

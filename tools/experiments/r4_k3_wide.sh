#!/bin/bash
# K3's wider-register instantiations (PGORB_QT_WIDE=1, default: 128 VGPRs at <= 4 waves per SIMD, 80 at 6) against the 64-VGPR one for every launch (=0)
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "quadtree or deep or roots or stages or randomised" 2>&1 | tail -1
run() { PGORB_QT_WIDE=$5 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap-leg --no-single-frame-leg --no-upload-leg --no-verify --width $1 --height $2 --features $3 --batch $4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['stage_ms_per_step']['quadtree'])"; }
echo "# K3 ms per step: width height features batch: wider instantiations where they apply (two runs) | 64 VGPRs everywhere (two runs)"
for cfg in "1920 1080 2000 1" "1920 1080 2000 32" "1920 1080 2000 128" "1920 1080 4000 128" "3840 2160 4000 32" "640 480 1000 64" "640 480 1000 512" "1280 720 1500 128" "1280 720 1000 256"; do set -- $cfg; echo "$1 $2 $3 $4: $(run $1 $2 $3 $4 1) $(run $1 $2 $3 $4 1) | $(run $1 $2 $3 $4 0) $(run $1 $2 $3 $4 0)"; done

#!/bin/bash
export TMPDIR=/tmp
for cfg in "1080p" "2160p --width 3840 --height 2160 --features 4000 --batch 32"; do
  set -- $cfg; tag=$1; shift
  rm -rf /tmp/qp; timeout 600 rocprofv3 --kernel-trace -d /tmp/qp -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-leg --no-single-frame-leg --no-upload-leg "$@" > /tmp/qp.log 2>&1
  db=$(find /tmp/qp -name '*.db' | head -1)
  echo "== $tag"; python tools/rocpd_summary.py stats "$db" | head -8
done
rm -rf /tmp/qp

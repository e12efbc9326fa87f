#!/bin/bash
export TMPDIR=/tmp
for nf in 2000 4000; do
  rm -rf /tmp/sp; timeout 600 rocprofv3 --kernel-trace -d /tmp/sp -- python tools/next_tier_bench.py --batch 128 --features $nf > /tmp/sp.log 2>&1
  db=$(find /tmp/sp -name '*.db' | head -1)
  echo "== features $nf"; python tools/rocpd_summary.py stats "$db" | grep -E "^kernel|k_sfi|k_search|k_proj|k_frame_grid|k_bow|k_feature|fillBuffer"
done
rm -rf /tmp/sp

#!/bin/bash
# kernel timeline of one step with pipeline_levels = $1 (rocprofv3 --kernel-trace), printed relative to the step's first kernel
export TMPDIR=/tmp
M=${1:-10}
D=/tmp/ptrace_$M; rm -rf $D
cat > /tmp/ptrace.py <<PY
import sys; sys.path.insert(0, ".")
import torch, pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
W,H,NF,B=1920,1080,2000,128
dev=torch.device("cuda",0)
ext=pg.ORBextractor(NF,1.2,8,20,7,max_width=W,max_height=H,max_batch=B,device=0)
frames=torch.from_numpy(synth_ride(0,W,H,B)).to(dev)
cap=ext.max_keypoints(W,H)
o=(torch.zeros((B,cap,7),dtype=torch.float32,device=dev),torch.zeros((B,cap,32),dtype=torch.uint8,device=dev),torch.zeros((B,),dtype=torch.int32,device=dev))
ext.set_option("pipeline_levels", $M)
for k in range(6): ext.extract_batch_device(frames,*o)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $D -- python /tmp/ptrace.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("$D/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_pyr", "k_fast", "k_quadtree", "k_describe"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last level-1 pyramid launch on
starts = [i for i, r in enumerate(rows) if "k_pyr" in r["Kernel_Name"]]
# find the beginning of the last run of k_pyr launches
i = starts[-1]
while i - 1 in starts: i -= 1
t0 = int(rows[i]["Start_Timestamp"])
for r in rows[i:]:
    print("%-28s q%-3s grid %-9s  %8.1f -> %8.1f us  (%.1f)" % (r["Kernel_Name"][:28], r.get("Queue_Id", "?"), r.get("Grid_Size", "?"),
          (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY

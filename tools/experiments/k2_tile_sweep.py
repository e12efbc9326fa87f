"""K2 tile sweep (BASELINE.json configs[2]): the block form's tile shape (cells per workgroup) against the
one-wave-per-cell form, on the bench workload.  Prints K2's time per step, algorithmic GB/s and the LDS per
workgroup; with PGORB_SWEEP_SHAPES="cx,cy;..." only those shapes.  Used by tools/experiments/k2_tile_sweep.sh
(which adds the rocprofv3 PMC traffic per shape)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride

W, H, NF, B = [int(x) for x in os.environ.get("PGORB_SWEEP_CFG", "1920,1080,2000,128").split(",")]
shapes = os.environ.get("PGORB_SWEEP_SHAPES", "0,0;1,1;2,1;2,2;4,1;4,2;3,3;4,4")
ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
fr = torch.from_numpy(synth_ride(0, W, H, B)).cuda()
inv = ext.GetInverseScaleFactors()
px = sum(int(np.rint(np.float32(W) * inv[l])) * int(np.rint(np.float32(H) * inv[l])) for l in range(8))
ref = None
for sh in shapes.split(";"):
    cx, cy = [int(v) for v in sh.split(",")]
    if cx == 0:
        ext.set_option("fast_kernel", 0)
    else:
        ext.set_option("fast_kernel", 1); ext.set_option("fast_block_cx", cx); ext.set_option("fast_block_cy", cy)
    for _ in range(2):
        k, d, n = ext.extract_batch_device(fr)
    ext.check_async(); torch.cuda.synchronize()
    ext.profile_begin(10)
    for _ in range(10):
        k, d, n = ext.extract_batch_device(fr)
    ncalls, ms = ext.profile_read()
    sig = (int(n.sum()), int(k[:, :100].view(torch.int32).sum()))
    if ref is None: ref = sig
    name = "one wave per cell (30 px)" if cx == 0 else "block %d x %d cells (%d x %d px)" % (cx, cy, 31 * cx, 31 * cy)
    print(json.dumps({"tile": name, "k2_ms": round(ms["fast"], 4), "algorithmic_GBps": round(px * B / ms["fast"] / 1e6, 1),
                      "frac_of_8TBps": round(px * B / ms["fast"] / 1e6 / 8000, 4), "same_output": sig == ref}))
ext.set_option("fast_kernel", 0)

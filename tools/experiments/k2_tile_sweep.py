"""K2 tile-shape sweep of the SHIPPED kernel (BASELINE.json configs[2]: "LDS tile-size sweep, rocprof HBM GB/s vs peak"): the LDS window
pitch (pgorb_set_option "fast_tile_pitch": 0 = 48 bytes with compile-time offsets, the shipped shape; 48 ... 128 = run-time pitch)
x waves per workgroup (1 | 4), on the bench workload.  Prints K2's HIP-event time per launch, algorithmic GB/s, LDS bytes per wave and
whether the output equals the shipped shape's.  PGORB_SWEEP_SHAPES="pitch,wpb;..." restricts the list (tools/experiments/k2_tile_sweep.sh
adds the rocprofv3 PMC traffic per shape)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride

W, H, NF, B = [int(x) for x in os.environ.get("PGORB_SWEEP_CFG", "1920,1080,2000,128").split(",")]
shapes = os.environ.get("PGORB_SWEEP_SHAPES", "0,1;48,1;64,1;80,1;96,1;128,1;0,4;64,4")
ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
fr = torch.from_numpy(synth_ride(0, W, H, B)).cuda()
inv = ext.GetInverseScaleFactors()
px = sum(int(np.rint(np.float32(W) * inv[l])) * int(np.rint(np.float32(H) * inv[l])) for l in range(8))
ref = None
for sh in shapes.split(";"):
    pitch, wpb = [int(v) for v in sh.split(",")]
    ext.set_option("fast_tile_pitch", pitch); ext.set_option("fast_waves_per_block", wpb)
    for _ in range(2):
        k, d, n = ext.extract_batch_device(fr)
    ext.check_async(); torch.cuda.synchronize()
    ext.profile_begin(10)
    for _ in range(10):
        k, d, n = ext.extract_batch_device(fr)
    ncalls, ms = ext.profile_read()
    nn = n.cpu().numpy()
    sig = (int(nn.sum()), bytes(k[0, :int(nn[0])].cpu().numpy().tobytes()), bytes(d[B - 1, :int(nn[B - 1])].cpu().numpy().tobytes()))
    if ref is None: ref = sig
    tp = pitch if pitch else 48
    lds = 42 * tp + max(36 * 40 + 1536, 2048) + 16            # window rows x pitch + score map + candidate list (fast.hip, 36-px cells)
    print(json.dumps({"tile": "window pitch %d B%s, %d wave(s) per workgroup" % (tp, "" if pitch else " (compile-time offsets: shipped)", wpb),
                      "lds_bytes_per_wave": lds, "k2_ms": round(ms["fast"], 4), "algorithmic_GBps": round(px * B / ms["fast"] / 1e6, 1),
                      "frac_of_8TBps": round(px * B / ms["fast"] / 1e6 / 8000, 4), "same_output": sig == ref}), flush=True)
ext.set_option("fast_tile_pitch", 0); ext.set_option("fast_waves_per_block", 1)

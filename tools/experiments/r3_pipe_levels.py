#!/usr/bin/env python3
"""K3 / K4-6 of one group of levels beside K2 of the next (pgorb_set_option "pipeline_levels"): step time per grouping,
outputs compared with the serial order.  usage: r3_pipe_levels.py [width height features batch]"""
import sys, time
sys.path.insert(0, ".")
import torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride

W, H, NF, B = (int(a) for a in (sys.argv[1:5] + ["1920", "1080", "2000", "128"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=0)
frames = torch.from_numpy(synth_ride(0, W, H, B)).to(dev)
cap = ext.max_keypoints(W, H)


def outs():
    return (torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
            torch.zeros((B,), dtype=torch.int32, device=dev))


pq = torch.arange(1, B, dtype=torch.int32, device=dev)
pt = torch.arange(0, B - 1, dtype=torch.int32, device=dev)
mout = (torch.empty((B - 1, cap), dtype=torch.int32, device=dev), torch.empty((B - 1, cap), dtype=torch.int16, device=dev),
        torch.empty((B - 1, cap), dtype=torch.int16, device=dev))


def run(mask, prio, o, seconds=1.0):
    ext.set_option("pipeline_levels", mask)

    def step():
        ext.extract_batch_device(frames, *o)
        ext.match_batch_device(o[1], o[2], pq, pt, mout)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                step()
            torch.cuda.synchronize(); k += 20
        best = min(best, (time.perf_counter() - t0) / k)
    ext.check_async()
    return best


import os
ext.set_option("pipeline_levels_priority", int(os.environ.get("PIPE_PRIO", "0")))
ref = outs()
t0 = run(0, 0, ref)
print("serial                      %.4f ms/step  %.0f frames/s" % (t0 * 1e3, B / t0))
nref = ref[2].cpu().tolist()
masks = [("L0 | 1-7", 0b10), ("L0 | 1-2 | 3-7", 0b1010), ("L0 | 1 | 2-3 | 4-7", 0b10110), ("0-1 | 2-7", 0b100), ("0-1 | 2-3 | 4-7", 0b10100),
         ("0-2 | 3-7", 0b1000), ("every level", 0b11111110), ("0 | 1 | 2 | 3-7", 0b1110), ("0-3 | 4-7", 0b10000), ("0 | 1-4 | 5-7", 0b100010), ("0-1 | 2-4 | 5-7", 0b100100), ("0 | 1 | 2-4 | 5-7", 0b100110)]
for name, m in masks:
    o = outs()
    t = run(m, 0, o)
    same = torch.equal(o[2], ref[2]) and all(torch.equal(o[1][f, :nref[f]], ref[1][f, :nref[f]]) and       # (keypoints as bits: class_id -1 is a NaN pattern)
                                             torch.equal(o[0][f, :nref[f]].view(torch.int32), ref[0][f, :nref[f]].view(torch.int32)) for f in range(B))
    print("%-26s  %.4f ms/step  %.0f frames/s  (%+.1f %%)  outputs equal: %s" % (name, t * 1e3, B / t, (t0 / t - 1) * 100, same))

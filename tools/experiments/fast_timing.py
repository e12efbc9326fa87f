"""Where a k_fast_cells wave spends its life (wall-clock ticks of 10 ns, summed over all waves).
Needs a developer build: make -C pilotguru_amd/csrc clean; make -C pilotguru_amd/csrc EXTRA=-DPGORB_FAST_TIMING"""
import sys, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd import _lib
from pilotguru_amd.synth import synth_ride
W, H, NF, B = 1920, 1080, 2000, 64
L = _lib.lib()
fn = L.pgorb_debug_fast_times
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
frames = torch.from_numpy(synth_ride(0, W, H, B)).cuda()
for it in range(3):
    ext.extract_batch_device(frames)
    torch.cuda.synchronize()
cells = sum(ext.level_cells(l) for l in range(8)) if hasattr(ext, "level_cells") else None
n = 1 << 20
log = np.zeros((n, 8), np.uint32)
fn(log.ctypes.data, n)
log = log[(log[:, 2] > 0) & (log[:, 3] > 0)]              # waves that ran a full cell
print("waves", len(log))
names = {0: "start -> cell record", 4: "-> addresses ready", 1: "-> window loads issued", 2: "-> window in LDS",
         5: "-> iniTh test + compaction", 6: "-> iniTh scores", 7: "-> iniTh NMS + emission", 3: "-> done (the minTh retry)"}
for i, nm in names.items():
    v = log[:, i] * 0.01
    print("   %-28s mean %6.2f us   median %6.2f   p90 %6.2f" % (nm, v.mean(), np.median(v), np.percentile(v, 90)))
print("   total mean %.2f us" % (log.sum(axis=1).mean() * 0.01))

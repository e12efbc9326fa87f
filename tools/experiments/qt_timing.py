"""Per-phase time of quadtree workgroup (frame 0, level 0).  Needs a developer build:
   make -C pilotguru_amd/csrc clean all EXTRA=-DPGORB_QT_TIMING"""
import sys, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd import _lib
from pilotguru_amd.synth import synth_ride
W, H, NF = 1920, 1080, 2000
L = _lib.lib()
fn = L.pgorb_debug_qt_times
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 24)()
for B in (1, 128):
    ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    frames = torch.from_numpy(synth_ride(0, W, H, B)).cuda()
    for it in range(3):
        ext.extract_batch_device(frames)
        torch.cuda.synchronize()
        fn(out, 1)
    t = [out[i] * 0.01 for i in range(10)]
    names = ["pyramid sums", "roots", "generations", "node map + best", "-", "-", "(barrier)", "select", "zeroing", "candidate pass"]
    print("batch", B, "ncand", out[10], "bfs gens", out[11], "sorted gens", out[12], "total us", round(sum(t), 1))
    for nm, v in zip(names, t): print("   %-16s %7.1f us" % (nm, v))
    print("   k_qt_leaves (frame 0, first group of level 0): level + band %.1f, loads issued %.1f, barrier %.1f, cell loop %.1f, barrier %.1f, flush %.1f us" % tuple(out[i] * 0.01 for i in range(16, 22)))

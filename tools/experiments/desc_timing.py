"""Where a k_describe wave spends its life (10 ns ticks).  Needs a developer build:
   make -C pilotguru_amd/csrc clean; make -C pilotguru_amd/csrc EXTRA=-DPGORB_DESC_TIMING"""
import sys, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd import _lib
from pilotguru_amd.synth import synth_ride
W, H, NF, B = 1920, 1080, 2000, 128
L = _lib.lib()
fn = L.pgorb_debug_desc_times
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
frames = torch.from_numpy(synth_ride(0, W, H, B)).cuda()
for it in range(3):
    ext.extract_batch_device(frames)
    torch.cuda.synchronize()
n = 1 << 19
log = np.zeros((n, 8), np.uint32)
fn(log.ctypes.data, n)
log = log[log[:, 6] > 0]
print("waves", len(log))
names = ["start -> selection record", "-> window in LDS", "-> moments reduced", "-> atan2", "-> row pass (MFMA) stored", "-> cos/sin", "-> 256 tests"]
for i, nm in enumerate(names):
    v = log[:, i] * 0.01
    print("   %-28s mean %6.2f us   median %6.2f   p90 %6.2f" % (nm, v.mean(), np.median(v), np.percentile(v, 90)))
print("   total mean %.2f us" % (log[:, :7].sum(axis=1).mean() * 0.01))

"""Phase times of K2's block form (developer build: make -C pilotguru_amd/csrc EXTRA=-DPGORB_FB_TIMING)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
B = 128
ext = pg.ORBextractor(2000, 1.2, 8, 20, 7, max_width=1920, max_height=1080, max_batch=B)
fr = torch.from_numpy(synth_ride(0, 1920, 1080, B)).cuda()
for _ in range(3):
    ext.extract_batch_device(fr)
torch.cuda.synchronize()
nb = 828
buf = np.zeros((nb, 16), np.uint32)
ext._L.pgorb_debug_fb_times(C.c_void_p(buf.ctypes.data), nb)
names = ["staged", "quick0", "score0", "nms0", "final0", "quick1", "score1", "nms1", "final1"]
t = buf[:, :9].astype(np.float64) * 10.0 / 1000.0      # us since block start
has1 = buf[:, 5] > 0
print("blocks", nb, "with pass 1:", has1.sum())
prev = np.zeros(nb)
for k, nm in enumerate(names):
    sel = np.ones(nb, bool) if k < 5 else has1
    if sel.sum() == 0: continue
    print("%-8s at %7.2f us  (phase %6.2f us)  n=%d" % (nm, t[sel, k].mean(), (t[sel, k] - prev[sel]).mean(), sel.sum()))
    prev = np.where(sel, t[:, k], prev)

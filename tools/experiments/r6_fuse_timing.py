"""Developer timing of the WORKGROUP-per-tile form of the fused launch (tools/experiments/r6_fused_cooperative_workgroup.hip.txt dropped in as
csrc/fused.hip with its host tables, built with make EXTRA=-DPGORB_FUSE_TIMING, loaded through PGORB_LIBRARY; the shipped one-wave-per-slot form has no
timers): the numbers of profiles/r06_fused_forms.txt -- per-wave ticks (10 ns) at the phase
borders of k_pyr_fast for level 0 of one 1080p batch: DMA issued / landed / barrier passed / resize done / cell 1 / cell 2 / cell 3."""
import ctypes as C
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
W, H, B = 1920, 1080, 128
ride = torch.from_numpy(synth_ride(1000, W, H, B)).cuda()
ext = pg.ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
for _ in range(3):
    ext.extract_batch_device(ride)
torch.cuda.synchronize()
L = ext._L
# the log holds the LAST launch that wrote each wave id: level 0 has the most workgroups, smaller levels overwrite the low ids of each frame
nw = 1 << 21
buf = np.zeros((nw, 8), np.uint32)
assert L.pgorb_debug_fuse_times(buf.ctypes.data_as(C.c_void_p), nw) == 0
gx0 = int(os.environ.get("GX0", "288"))
allw = buf[:B * gx0 * 4]
alive = allw[allw[:, 4] > 0]
t0 = alive[:, 7].astype(np.int64); t1 = t0 + alive[:, 4]
span = (t1.max() - t0.min()) * 0.01
print("level 0 launch: %d waves logged, span %.1f us, mean wave life %.2f us -> mean waves alive %.0f (%.1f per CU)" %
      (len(alive), span, alive[:, 4].mean() * 0.01, alive[:, 4].sum() * 0.01 / span, alive[:, 4].sum() * 0.01 / span / 256))
raw = allw[64 * gx0 * 4: 65 * gx0 * 4]                    # frame 64: the middle of the launch
t = raw.astype(np.float64) * 0.01      # us
names = ["dma issued", "dma landed", "barrier", "first item", "all items"]
for k, nme in enumerate(names):
    col = t[:, k]; col = col[col > 0]
    if len(col):
        print("%-12s n=%5d mean %.2f us  p50 %.2f  p90 %.2f  max %.2f" % (nme, len(col), col.mean(), np.median(col), np.percentile(col, 90), col.max()))
print("items per wave: mean %.2f (cells %.2f)" % (raw[:, 5].mean(), raw[:, 6].mean()))
last = t[:, 4].reshape(-1, 4)
last = last[last.max(axis=1) > 0]
print("wave end: mean %.2f us; workgroup end (max over its 4 waves): mean %.2f; mean/max over waves %.2f" %
      (last.mean(), last.max(axis=1).mean(), (last.mean(axis=1) / last.max(axis=1)).mean()))

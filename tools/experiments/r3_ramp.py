"""How the step time settles after a cold start: consecutive groups of 5 steps, each timed with HIP events (1080p / 2000, batch 128)."""
import sys, time
sys.path.insert(0, ".")
import torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
W, H, NF, B = 1920, 1080, 2000, 128
dev = torch.device("cuda", 0)
ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=0)
frames = torch.from_numpy(synth_ride(0, W, H, B)).to(dev)
cap = ext.max_keypoints(W, H)
o = (torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev))
pq = torch.arange(1, B, dtype=torch.int32, device=dev); pt = torch.arange(0, B - 1, dtype=torch.int32, device=dev)
mout = (torch.empty((B - 1, cap), dtype=torch.int32, device=dev), torch.empty((B - 1, cap), dtype=torch.int16, device=dev), torch.empty((B - 1, cap), dtype=torch.int16, device=dev))
def step():
    ext.extract_batch_device(frames, *o); ext.match_batch_device(o[1], o[2], pq, pt, mout)
step(); torch.cuda.synchronize()            # plan, arenas
time.sleep(2.0)                             # let the clocks fall back
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for g in range(40):
    for _ in range(5): step()
    evs[g + 1].record()
torch.cuda.synchronize()
ms = [evs[g].elapsed_time(evs[g + 1]) / 5 for g in range(40)]
print("ms per step, groups of 5 steps after 2 s idle:", " ".join("%.3f" % m for m in ms))

"""Does a HIP graph of the bench step beat direct launches?  The step (K1-K7: ~16 launches + 2 memsets on one stream) is captured with
torch.cuda.graph on torch's capture stream and replayed; same buffers, same work.  usage: python tools/experiments/graph_step.py [w h nf batch]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
w, h, nf, B = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (1920, 1080, 2000, 128)
ride = synth_ride(0, w, h, min(B, 32))
fr = torch.from_numpy(np.concatenate([ride] * (B // len(ride)))).cuda()
ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
cap = ext.max_keypoints(w, h)
kps = torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"); desc = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda")
n = torch.empty((B,), dtype=torch.int32, device="cuda")
pq = torch.arange(1, B, dtype=torch.int32, device="cuda"); pt = torch.arange(0, B - 1, dtype=torch.int32, device="cuda")
mout = None


def step(stream=None):
    global mout
    ext.extract_batch_device(fr, kps, desc, n, stream=stream)
    mout = ext.match_batch_device(desc, n, pq, pt, mout, stream=stream)


for _ in range(5): step()
torch.cuda.synchronize()


def timed(fn, reps=40):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


direct = timed(step)
ref = [t.clone() for t in (kps, desc, n) + tuple(mout)]
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(s.cuda_stream); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        step(s.cuda_stream)
torch.cuda.synchronize()
graph = timed(g.replay)
same = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(ref, (kps, desc, n) + tuple(mout)))
print("%dx%d / %d, batch %d: direct %.4f ms per step (%.0f frames/s), graph replay %.4f ms (%.0f frames/s), outputs equal: %s" % (
    w, h, nf, B, direct, B / direct * 1e3, graph, B / graph * 1e3, same))

"""Single-frame latency: direct launches vs one hipGraph replay of extract + match (the library's
device-side entry points are capture-safe after the first, allocating call)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
W, H, NF, B = 1920, 1080, 2000, 2
ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
frames = torch.from_numpy(synth_ride(0, W, H, B)).cuda()
cap = ext.max_keypoints(W, H)
kps = torch.empty((B, cap, 7), dtype=torch.float32, device='cuda')
desc = torch.empty((B, cap, 32), dtype=torch.uint8, device='cuda')
n = torch.empty((B,), dtype=torch.int32, device='cuda')
pq = torch.tensor([1], dtype=torch.int32, device='cuda'); pt = torch.tensor([0], dtype=torch.int32, device='cuda')
mout = (torch.empty((1, cap), dtype=torch.int32, device='cuda'), torch.empty((1, cap), dtype=torch.int16, device='cuda'),
        torch.empty((1, cap), dtype=torch.int16, device='cuda'))
s = torch.cuda.Stream()
def step(stream):
    ext.extract_batch_device(frames[1:2], kps[1:2], desc[1:2], n[1:2], stream=stream.cuda_stream)
    ext.match_batch_device(desc, n, pq, pt, mout, stream=stream.cuda_stream)
with torch.cuda.stream(s):
    ext.extract_batch_device(frames[0:1], kps[0:1], desc[0:1], n[0:1], stream=s.cuda_stream)
    for _ in range(3): step(s)
    s.synchronize()
    ref = (kps[1].clone(), desc[1].clone(), n.clone(), mout[0].clone())
    def timeit(f, k=200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(k): f()
        e1.record(s); s.synchronize()
        return e0.elapsed_time(e1) / k * 1e3
    t_direct = timeit(lambda: step(s))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step(s)
    g.replay(); s.synchronize()
    ok = torch.equal(kps[1].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(desc[1], ref[1]) and torch.equal(mout[0], ref[3])
    t_graph = timeit(lambda: g.replay())
print("direct launches: %.1f us per frame   graph replay: %.1f us per frame   results identical: %s" % (t_direct, t_graph, ok))

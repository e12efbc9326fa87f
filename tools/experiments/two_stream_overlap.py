"""Experiment: does running two half-batches on two HIP streams (two contexts) overlap kernels with different
bottlenecks (K1 HBM, K2 VALU issue, K3 latency, K7 matrix pipe)?  Compares one context x 128 frames with
two contexts x 64 frames on two streams, and with 4 x 32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride

W, H, NF, B = 1920, 1080, 2000, 128
ride = torch.from_numpy(synth_ride(0, W, H, B)).cuda()
ride = torch.cat([ride, ride, ride, ride])
for parts, bs in ((1, 128), (2, 64), (4, 32), (2, 128), (3, 128), (2, 256)):
    exts = [pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=bs) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cap = exts[0].max_keypoints(W, H)
    outs = [(torch.empty((bs, cap, 7), dtype=torch.float32, device="cuda"), torch.empty((bs, cap, 32), dtype=torch.uint8, device="cuda"),
             torch.empty((bs,), dtype=torch.int32, device="cuda")) for _ in range(parts)]
    pq = torch.arange(1, bs, dtype=torch.int32, device="cuda"); pt = torch.arange(0, bs - 1, dtype=torch.int32, device="cuda")
    mouts = [(torch.empty((bs - 1, cap), dtype=torch.int32, device="cuda"), torch.empty((bs - 1, cap), dtype=torch.int16, device="cuda"),
              torch.empty((bs - 1, cap), dtype=torch.int16, device="cuda")) for _ in range(parts)]
    def step():
        for i in range(parts):
            s = streams[i].cuda_stream
            exts[i].extract_batch_device(ride[i * bs:(i + 1) * bs], *outs[i], stream=s)
            exts[i].match_batch_device(outs[i][1], outs[i][2], pq, pt, mouts[i], stream=s)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%d context(s) x %d frames on %d stream(s): %.3f ms per %d frames -> %.0f frames/s" % (parts, bs, parts, dt * 1e3, parts * bs, parts * bs / dt))
    del exts

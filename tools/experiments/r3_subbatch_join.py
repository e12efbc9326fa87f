"""Experiment: one batch of 128 frames as K sub-batches on K HIP streams WITH a device-side join per step (what a library-internal
split would have to do: results of a call are ordered on the caller's stream), against the unsplit batch and the free-running split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride

W, H, NF, B = 1920, 1080, 2000, 128
ride = torch.from_numpy(synth_ride(0, W, H, B)).cuda()
main = torch.cuda.Stream()
for parts, join in ((1, False), (2, False), (2, True), (4, False), (4, True)):
    bs = B // parts
    exts = [pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=bs) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cap = exts[0].max_keypoints(W, H)
    kps = torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"); desc = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.empty((B,), dtype=torch.int32, device="cuda")
    pq = torch.arange(1, B, dtype=torch.int32, device="cuda"); pt = torch.arange(0, B - 1, dtype=torch.int32, device="cuda")
    mout = (torch.empty((B - 1, cap), dtype=torch.int32, device="cuda"), torch.empty((B - 1, cap), dtype=torch.int16, device="cuda"),
            torch.empty((B - 1, cap), dtype=torch.int16, device="cuda"))
    def step():
        if parts == 1:
            exts[0].extract_batch_device(ride, kps, desc, n, stream=main.cuda_stream)
        else:
            ev0 = torch.cuda.Event(); ev0.record(main)
            for i in range(parts):
                if join: streams[i].wait_event(ev0)
                exts[i].extract_batch_device(ride[i * bs:(i + 1) * bs], kps[i * bs:(i + 1) * bs], desc[i * bs:(i + 1) * bs], n[i * bs:(i + 1) * bs], stream=streams[i].cuda_stream)
                e = torch.cuda.Event(); e.record(streams[i]); main.wait_event(e)
        exts[0].match_batch_device(desc, n, pq, pt, mout, stream=main.cuda_stream)       # all 127 pairs, after every sub-batch
    for _ in range(3): step()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40): step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40)
    print("%d sub-batch(es) of %d, join per step %-5s: %.3f ms per 128 frames -> %.0f frames/s" % (parts, bs, join, best * 1e3, B / best))
    del exts

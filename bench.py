#!/usr/bin/env python3
"""bench.py -- frames/s of the ORB front end (extract + match) on MI355X.

One "step" = one pass of the hot path over one batch of B synthetic 1920x1080 grey frames
that are already resident in HBM: K1 pyramid -> K2 per-cell FAST -> K3 quadtree ->
K4-6 orientation+blur+rBRIEF (2000 keypoints/frame) -> K7 best-2 Hamming match of every frame
against its predecessor in the ride.  Workload = BASELINE.json configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: frames are independent, so every rank runs its own ride on its own GPU (weak
scaling, no data-path collective); the only collective is one RCCL broadcast of the ORB
vocabulary at start-up (outside the timed region), plus the barrier / max-over-ranks timing.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def level_sizes(ext, w, h):
    inv = ext.GetInverseScaleFactors()
    import numpy as np
    return [(int(np.rint(np.float32(w) * inv[l])), int(np.rint(np.float32(h) * inv[l])))
            for l in range(ext.nlevels)]


def algorithmic_bytes(sizes, nkp):
    """Per-frame algorithmic bytes of each kernel group."""
    px = [a * b for a, b in sizes]
    return {
        "pyramid": sum(px[:-1]) + sum(px[1:]),          # reads of levels 0..L-2 + writes of 1..L-1
        "fast": sum(px),                                # one detection read of every level
        "quadtree": 0,
        "describe": nkp * (43 * 43 + 60),               # raw window + keypoint/descriptor out
        "match": 2 * nkp * 32 + nkp * 8,
    }


_CPU_WORKER = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from oracle import orb_oracle
from pilotguru_amd.synth import synth_ride
w, h, nf, budget, seed = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
ride = synth_ride(1000 + seed, w, h, 4)
ora = orb_oracle.OrbOracle(nf, 1.2, 8, 20, 7)
kp, prev = ora.extract(ride[0])                       # warm-up, also the first "previous frame"
sys.stdout.write("ready\n"); sys.stdout.flush()
sys.stdin.readline()                                   # start gate
t0 = time.perf_counter(); done = 0
while time.perf_counter() - t0 < budget:
    kp, desc = ora.extract(ride[(done + 1) % 4])
    orb_oracle.hamming_best2(desc, prev)
    prev = desc; done += 1
print(done, time.perf_counter() - t0)
"""


def cpu_baseline(w, h, nfeatures, budget_s=12.0):
    """The CPU oracle (oracle/: a port of the reference path, the reference binary itself cannot
    be built here) timed on this host: one extractor per core like running one ride per core
    (the reference runs extraction on one thread, Frame.cc:254), on a bounded sample of the same
    workload: extract + best-2 match against the previous frame, ~12 s per worker."""
    import subprocess
    cores = min(os.cpu_count() or 1, 64)
    procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, ROOT, str(w), str(h), str(nfeatures), str(budget_s), str(i)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, universal_newlines=True)
             for i in range(cores)]
    for p in procs:
        p.stdout.readline()                            # all workers generated their frames
    for p in procs:
        p.stdin.write("go\n"); p.stdin.flush()
    frames, tmax, single = 0, 0.0, 0.0
    for i, p in enumerate(procs):
        out = p.stdout.readline().split()
        p.wait()
        d, t = int(out[0]), float(out[1])
        frames += d; tmax = max(tmax, t)
    return {"value": frames / tmax, "unit": "frames/s", "cores": cores, "kind": "port",
            "per_core": frames / tmax / cores,
            "sample": "%d synthetic %dx%d frames over %d worker processes (one oracle extractor per core), "
                      "%d features, extract + best-2 match vs previous frame, %.0f s budget"
                      % (frames, w, h, cores, nfeatures, budget_s)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import pilotguru_amd as pg
    from pilotguru_amd.synth import synth_ride

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("PGORB_BENCH_FORCE_DIST"):    # the env switch exercises the N>1 code on one GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    W, H, B, NF = args.width, args.height, args.batch, args.features
    ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=local_rank)

    # one ride per rank (ride id = rank): B consecutive frames, resident in HBM
    from pilotguru_amd import dist as pgd0
    ride = synth_ride(pgd0.ride_for_rank(rank, world)[0], W, H, B)
    frames = torch.from_numpy(ride).to(dev)
    cap = ext.max_keypoints(W, H)
    kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    n = torch.empty((B,), dtype=torch.int32, device=dev)
    pq = torch.arange(1, B, dtype=torch.int32, device=dev)      # frame f (query) vs f-1 (train)
    pt = torch.arange(0, B - 1, dtype=torch.int32, device=dev)
    mout = (torch.empty((B - 1, cap), dtype=torch.int32, device=dev),
            torch.empty((B - 1, cap), dtype=torch.int16, device=dev),
            torch.empty((B - 1, cap), dtype=torch.int16, device=dev))

    # the one collective of this path: broadcast the (synthetic) ORB vocabulary root -> peers
    vocab_bytes = 0
    if dist is not None:
        from pilotguru_amd import dist as pgd
        from pilotguru_amd.vocab import synth_vocabulary_blob
        blob = synth_vocabulary_blob(k=10, L=5, seed=7) if rank == 0 else None
        vocab = pgd.broadcast_vocabulary(blob, 0, dev)
        vocab_bytes = int(vocab.numel())
        # every rank makes the received blob resident in its context and transforms the same
        # probe descriptors; the word ids must agree across ranks (config 4's criterion)
        import ctypes as C
        ext._check(ext._L.pgorb_vocab_upload_device(ext._h, C.c_void_p(vocab.data_ptr()), vocab.numel(),
                                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        g = torch.Generator().manual_seed(1234)
        probe = torch.randint(0, 256, (2048, 32), generator=g, dtype=torch.uint8).to(dev)
        pw = torch.empty(2048, dtype=torch.int32, device=dev)
        pwt = torch.empty(2048, dtype=torch.float64, device=dev)
        pn = torch.empty(2048, dtype=torch.int32, device=dev)
        ext._check(ext._L.pgorb_bow_transform_device(
            ext._h, C.c_void_p(probe.data_ptr()), 2048, 4, C.c_void_p(pw.data_ptr()), C.c_void_p(pwt.data_ptr()),
            C.c_void_p(pn.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        sig = torch.stack([pw.to(torch.int64).sum(), (pw.to(torch.int64) * torch.arange(2048, device=dev)).sum(),
                           pn.to(torch.int64).sum()])
        sigs = [torch.empty_like(sig) for _ in range(world)]
        dist.all_gather(sigs, sig)
        if not all(torch.equal(sigs[0], x) for x in sigs):
            raise SystemExit("BoW words differ across ranks after the vocabulary broadcast")

    def step():
        ext.extract_batch_device(frames, kps, desc, n)
        ext.match_batch_device(desc, n, pq, pt, mout)

    for _ in range(args.warmup):
        step()
    ext.check_async()
    torch.cuda.synchronize()
    counts = n.cpu().numpy()

    ext.profile_begin(args.steps)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        from pilotguru_amd import dist as pgd
        elapsed = pgd.max_over_ranks(elapsed, dev)
    ncalls, stage_ms = ext.profile_read()
    ext.check_async()

    if rank == 0:
        frames_total = world * B * args.steps
        fps = frames_total / elapsed
        sizes = level_sizes(ext, W, H)
        nkp = float(counts.mean())
        abytes = algorithmic_bytes(sizes, nkp)
        dom = max(("pyramid", "fast", "quadtree", "describe", "match"), key=lambda s: stage_ms[s])
        # the roofline object describes the dominant HBM-streaming kernel; the quadtree moves
        # no image bytes, so when it dominates wall time the image kernel with most time is used
        cands = [s for s in ("pyramid", "fast", "describe", "match")]
        rk = max(cands, key=lambda s: stage_ms[s])
        launches = 7 if rk == "pyramid" else 1
        ach = (abytes[rk] * B / launches) / (stage_ms[rk] / launches * 1e-3) / 1e9
        kname = {"pyramid": "k_pyr_resize_rows4_lds", "fast": "k_fast_cells", "describe": "k_describe",
                 "match": "k_match_mfma"}[rk]
        traffic = None                                  # HBM bytes per launch from the committed PMC passes
        try:
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):   # newest round first
                tr = json.load(open(path))
                if tr["workload"] == {"width": W, "height": H, "features": NF, "batch": B} and kname in tr["kernels"]:
                    k = tr["kernels"][kname]
                    traffic = (2.0 * k["fetch_kb"] + k["write_kb"]) * 1024.0
                    break
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "frames/sec ORB extract+match, 1080p @ 2000 kp/frame",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d grayscale, %d kp/frame, 8-level pyramid, batch %d frames/GPU "
                                   "resident in HBM, extract + best-2 Hamming match vs previous frame"
                                   % (W, H, NF, B),
                       "batch": B, "keypoints_per_frame": nkp, "parallelism": "frames-sharded x%d" % world,
                       "vocab_broadcast_bytes": vocab_bytes},
            "roofline": {"bound": "hbm", "kernel": kname,
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": abytes[rk] * B / launches,
                         "launch_ms": stage_ms[rk] / launches},
            "stage_ms_per_step": stage_ms, "dominant_stage": dom,
            "whole_path_algorithmic_GBps": sum(abytes.values()) * fps / world / 1e9,
        }
        if not args.no_cpu_baseline and world == 1:            # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(W, H, NF)
            out["speedup_vs_cpu_all_cores"] = fps / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_per_core"] = fps / out["cpu_baseline"]["per_core"]
        line = json.dumps(out)
    else:
        line = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # RCCL prints its version banner through C stdio, which a pipe only sees at exit: flush it
        # first so that the JSON line is the LAST line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()

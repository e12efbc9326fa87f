"""Multi-GPU plumbing: one process per GPU, frames sharded by ride, ONE collective.

The reference is a single process.  Extraction of a frame depends on no
other frame, so the path shards by independent units: ride r -> rank r (config 4) with no
data-path collective.  The only exchange is the start-up broadcast of the ORB vocabulary the
reference loads once and shares by pointer (src/optical_trajectories.cc:87-94): root -> peers
with torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).
"""
import os


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def ride_for_rank(rank, world, nrides=None):
    """Ride ids a rank owns: ride r -> rank r mod world (nrides defaults to one per rank)."""
    nrides = world if nrides is None else nrides
    return [r for r in range(nrides) if r % world == rank]


def broadcast_vocabulary(blob, root, device):
    """Broadcast a packed vocabulary (uint8 tensor, pilotguru_amd.vocab) from `root`.

    Two messages: the byte count, then the blob -- one flat message of a few MB..50 MB, i.e. a
    1-hop broadcast bound by a single xGMI link (~153 GB/s) when the backend is RCCL.  Returns
    the blob on `device` on every rank."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    n = torch.tensor([blob.numel() if rank == root else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, root)
    if rank == root:
        t = blob.to(device)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(t, root)
    return t


def max_over_ranks(seconds, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def frame_chunk_for_rank(nframes, rank, world):
    """One ride over several GPUs: contiguous chunks of frames with a one-frame overlap.

    Returns (first_extracted, first_owned, stop): the rank extracts frames [first_extracted, stop) and owns frame f
    -- its keypoints and its match against frame f-1 -- for f in [first_owned, stop).  first_extracted is
    first_owned - 1 except for the first chunk: the border frame is extracted twice, nothing is exchanged."""
    per, extra = divmod(nframes, world)
    first = rank * per + min(rank, extra)
    stop = first + per + (1 if rank < extra else 0)
    return (max(first - 1, 0) if stop > first else first), first, stop


def window_range_for_rank(nwindows, rank, world):
    """fit_motion's sliding windows are independent fits: contiguous, balanced ranges, no exchange but the gather."""
    per, extra = divmod(nwindows, world)
    first = rank * per + min(rank, extra)
    return first, first + per + (1 if rank < extra else 0)


def fit_velocity_windows_sharded(ctx, gps, rotations, accelerations, locations_batch_size=40, locations_shift_step=5,
                                 optimization_iters=500, fit=None):
    """pilotguru_amd.calibration.FitVelocityWindows with the windows split over the ranks of the default group.

    Window w only looks at GPS fixes [w * step, w * step + batch) (src/fit_motion.cc:173-183), so a rank fits the
    windows of its range on the slice of the GPS series they touch and every rank ends up with all results
    (one all_gather of a few KB).  `fit` defaults to the GPU path; tests pass a CPU checker."""
    import numpy as np
    import torch.distributed as dist
    if fit is None:
        from .calibration import FitVelocityWindows as fit
    v, t = np.asarray(gps[0]), np.asarray(gps[1])
    n, s, b = len(v), int(locations_shift_step), int(locations_batch_size)
    nw = (n + s - 1) // s if n > 0 else 0
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    a, e = window_range_for_rank(nw, rank, world)
    if e > a:
        lo, hi = a * s, min(n, (e - 1) * s + b)
        x, res, it = fit(ctx, (v[lo:hi], t[lo:hi]), rotations, accelerations, b, s, optimization_iters)
        mine = (np.asarray(x)[:e - a], np.asarray(res)[:e - a], np.asarray(it)[:e - a])
    else:
        mine = (np.zeros((0, 9)), np.zeros(0), np.zeros(0, np.int32))
    parts = [mine]
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, mine)
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))

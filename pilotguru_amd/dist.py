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

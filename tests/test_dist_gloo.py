"""world_size-2 gloo test of the N>1 path bench.py uses: ride sharding, the single vocabulary
broadcast, and max-over-ranks timing.  Runs on CPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pilotguru_amd import dist as pgd
    from pilotguru_amd.vocab import synth_vocabulary_blob, unpack_vocabulary
    dev = torch.device("cpu")
    blob = synth_vocabulary_blob(k=3, L=3, seed=5) if rank == 0 else None
    got = pgd.broadcast_vocabulary(blob, 0, dev)
    u = unpack_vocabulary(got.numpy())
    t = pgd.max_over_ranks(1.0 + rank, dev)
    rides = pgd.ride_for_rank(rank, world, nrides=5)
    out.put((rank, int(got.numel()), int(np.frombuffer(got.numpy().tobytes(), np.uint8).astype(np.uint64).sum()),
             u["nnodes"], t, rides))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_vocab_broadcast_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, n0, s0, nodes0, t0, rides0), (r1, n1, s1, nodes1, t1, rides1) = res
    assert (n0, s0, nodes0) == (n1, s1, nodes1) and nodes0 == 1 + 3 + 9 + 27
    assert t0 == t1 == 2.0                                   # MAX over ranks
    assert rides0 == [0, 2, 4] and rides1 == [1, 3]          # every ride owned exactly once

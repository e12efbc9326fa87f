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
    # fit_motion's windows split over the ranks; the CPU oracle stands in for the GPU fit
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_calibration import imu_ride
    from oracle import orb_oracle
    gps, rot, acc = imu_ride(8, n_gps=23)
    cpu_fit = lambda ctx, g, r, a, b, s, it: orb_oracle.fit_windows(*g, *r, *a, b, s, it)
    sx, sres, sit = pgd.fit_velocity_windows_sharded(None, gps, rot, acc, 8, 3, 25, fit=cpu_fit)
    ox, ores, oit = orb_oracle.fit_windows(*gps, *rot, *acc, 8, 3, 25)
    assert np.array_equal(sx.view(np.uint64), ox.view(np.uint64)) and np.array_equal(sres.view(np.uint64), ores.view(np.uint64))
    assert np.array_equal(sit, oit) and len(sit) == 8
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


def test_frame_chunks_and_window_ranges_cover_everything_once():
    from pilotguru_amd import dist as pgd
    for world in (1, 2, 3, 8):
        for n in (0, 1, 2, 7, 8, 9, 100, 1001):
            owned, prev_stop = [], 0
            for r in range(world):
                fe, fo, stop = pgd.frame_chunk_for_rank(n, r, world)
                assert fo == prev_stop and fe == (max(fo - 1, 0) if stop > fo else fo)      # contiguous, one frame of overlap
                owned += list(range(fo, stop)); prev_stop = stop
            assert owned == list(range(n))                                               # every frame (and pair f-1, f) exactly once
            w = [pgd.window_range_for_rank(n, r, world) for r in range(world)]
            assert w[0][0] == 0 and w[-1][1] == n and all(w[i][1] == w[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in w) - min(b - a for a, b in w) <= 1


def _id_failure_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pilotguru_amd import _lib
    from pilotguru_amd import dist as pgd

    class _NoRccl:                                             # libpgorb with a librccl that cannot make an id (rank 0 is the one that asks)
        def __init__(self, real): self._real = real
        def __getattr__(self, name): return getattr(self._real, name)
        def pgorb_comm_unique_id(self, ident): return -6
    _lib._lib = _NoRccl(_lib.lib())
    raised = False
    try:
        pgd.VocabularyComm.from_torch_group(None)
    except RuntimeError as e:
        raised = "pgorb_comm_unique_id failed on rank 0" in str(e)
    # every rank left the id broadcast: the NEXT collective (bench.py's all_reduce of the fallback flag) pairs up on both
    flag = torch.tensor([0 if raised else 1], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    out.put((rank, raised, int(flag.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_id_handover_failure_is_raised_on_every_rank_together():
    """rank 0 used to raise BEFORE the id broadcast when pgorb_comm_unique_id failed, leaving its peers blocked in that
    broadcast while it went on to the caller's next collective.  Now (ok, id) travels in the broadcast and every rank raises after it."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_id_failure_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, 0), (1, True, 0)]

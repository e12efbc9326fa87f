"""DBoW2 vocabulary path: text loader, blob, BowVector/FeatureVector accumulation (pinned
against the REAL reference BowVector.cpp/FeatureVector.cpp via oracle/_ref), and the GPU
tree descent against the oracle."""
import os

import numpy as np
import pytest

from pilotguru_amd import vocab as V


def _vocab_file(tmp_path, k=4, L=3, seed=1, trailing_newline=False):
    desc, weight, parent = V.synth_vocabulary(k, L, seed)
    # a few stop words (weight 0) so the "w > 0" branch is exercised
    weight[-3:] = 0.0
    path = os.path.join(str(tmp_path), "voc.txt")
    V.write_vocabulary_text(path, k, L, desc, weight, parent, trailing_newline=trailing_newline)
    return path, desc, weight, parent


def test_text_loader_matches_oracle_and_python_pack(tmp_path, oracle):
    path, desc, weight, parent = _vocab_file(tmp_path, trailing_newline=True)   # trailing newline is skipped
    voc = V.ORBVocabulary(text_file=path)
    ora = oracle.VocabOracle(path)
    assert (voc.k, voc.L, voc.nnodes, voc.nwords) == (ora.k, ora.L, ora.nnodes, ora.nwords) == (4, 3, 85, 64)
    assert np.array_equal(voc.blob(), V.pack_vocabulary(4, 3, desc, weight, parent))
    # blob round trip through the C ABI
    voc2 = V.ORBVocabulary(blob=voc.blob())
    assert np.array_equal(voc2.blob(), voc.blob())
    with pytest.raises(ValueError):
        V.ORBVocabulary(text_file=os.path.join(str(tmp_path), "missing.txt"))
    with pytest.raises(ValueError):
        V.ORBVocabulary(blob=np.zeros(128, np.uint8))


def test_vocabulary_cache_beside_the_text_file(tmp_path):
    """pgorb_vocab_load_cached: the first load parses the text and writes `<file>.pgvoc`, the second takes the cache (same blob);
    a changed text file (size / mtime) or a damaged cache falls back to the text and rewrites the cache."""
    path = _vocab_file(tmp_path, 5, 3, seed=3)[0]
    ref = V.ORBVocabulary(text_file=path).blob()
    a = V.ORBVocabulary(text_file=path, cache=True)
    assert not a.from_cache and os.path.exists(path + ".pgvoc") and np.array_equal(a.blob(), ref)
    b = V.ORBVocabulary(text_file=path, cache=True)
    assert b.from_cache and np.array_equal(b.blob(), ref)
    # another vocabulary under the same name: the cache is stale
    path2 = _vocab_file(tmp_path, 4, 3, seed=9)[0]
    assert path2 == path
    os.utime(path, ns=(1_700_000_000_000_000_000, 1_700_000_000_123_456_789))
    ref2 = V.ORBVocabulary(text_file=path).blob()
    c = V.ORBVocabulary(text_file=path, cache=True)
    assert not c.from_cache and np.array_equal(c.blob(), ref2) and not np.array_equal(ref2[:64], ref[:64])
    assert V.ORBVocabulary(text_file=path, cache=True).from_cache
    # a damaged cache (truncated; flipped structure bytes) is ignored
    raw = open(path + ".pgvoc", "rb").read()
    open(path + ".pgvoc", "wb").write(raw[:len(raw) // 2])
    d = V.ORBVocabulary(text_file=path, cache=True)
    assert not d.from_cache and np.array_equal(d.blob(), ref2)
    bad = bytearray(open(path + ".pgvoc", "rb").read()); bad[32 + 16] ^= 0xFF      # the blob header's node count
    open(path + ".pgvoc", "wb").write(bytes(bad))
    e = V.ORBVocabulary(text_file=path, cache=True)
    assert not e.from_cache and np.array_equal(e.blob(), ref2)
    with pytest.raises(ValueError):
        V.ORBVocabulary(text_file=os.path.join(str(tmp_path), "missing.txt"), cache=True)


def test_bow_vectors_match_real_reference_code(oracle):
    """pgorb_bow_vectors (product, host) == reference BowVector::addWeight/normalize and
    FeatureVector::addFeature, bit for bit (oracle/_ref built from /root/reference)."""
    if oracle.ref_dbow2() is None:
        pytest.skip("oracle/_ref/libdbow2_ref.so not built and /root/reference absent")
    rng = np.random.RandomState(3)
    for n in (0, 1, 7, 500, 2000):
        word = rng.randint(0, 300, n).astype(np.uint32)
        wtab = np.round(rng.uniform(0.0, 9.0, 300), 5)
        wtab[rng.randint(0, 300, 20)] = 0.0                   # stop words
        weight = wtab[word]
        node = (word // 7).astype(np.uint32)
        got = V.bow_vectors(word, weight, node, scoring=0, weighting=0)
        ref = oracle.ref_bow_vectors(word, weight, node)
        for g, r in zip(got[0] + got[1], ref[0] + ref[1]):
            assert g.dtype == r.dtype and g.tobytes() == r.tobytes()
        if n:
            assert abs(got[0][1].sum() - 1.0) < 1e-12           # L1 normalised


@pytest.mark.parametrize("scoring", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("weighting", [0, 1, 2, 3])
def test_bow_vectors_every_scoring_and_weighting_type(oracle, tmp_path, scoring, weighting):
    """ScoringObject.h:74-89: L1 norm for L1_NORM / CHI_SQUARE / KL / BHATTACHARYYA, L2 for L2_NORM, none
    for DOT_PRODUCT (then tf / |v| for TF weightings, TemplatedVocabulary.h:1164-1170).  Product ==
    reference BowVector code == oracle, bit for bit."""
    if oracle.ref_dbow2() is None:
        pytest.skip("oracle/_ref/libdbow2_ref.so not built and /root/reference absent")
    rng = np.random.RandomState(40 + scoring * 4 + weighting)
    n = 700
    word = rng.randint(0, 120, n).astype(np.uint32)
    wtab = np.round(rng.uniform(0.0, 9.0, 120), 5)
    wtab[rng.randint(0, 120, 9)] = 0.0
    weight, node = wtab[word], (word // 5).astype(np.uint32)
    got = V.bow_vectors(word, weight, node, scoring=scoring, weighting=weighting)
    ref = oracle.ref_bow_vectors(word, weight, node, scoring, weighting)
    for g, r in zip(got[0] + got[1], ref[0] + ref[1]):
        assert g.dtype == r.dtype and g.tobytes() == r.tobytes()
    val = got[0][1]
    if scoring == 1:
        assert abs(np.sqrt((val * val).sum()) - 1.0) < 1e-12
    elif scoring != 5:
        assert abs(np.abs(val).sum() - 1.0) < 1e-12
    # the oracle's whole transform() with this header agrees as well
    desc, wgt, parent = V.synth_vocabulary(4, 3, 1)
    path = os.path.join(str(tmp_path), "voc_%d_%d.txt" % (scoring, weighting))
    V.write_vocabulary_text(path, 4, 3, desc, wgt, parent, scoring=scoring, weighting=weighting)
    ora = oracle.VocabOracle(path)
    feats = rng.randint(0, 256, (200, 32)).astype(np.uint8)
    (bid, bval), ofv = ora.transform(feats, levelsup=2)
    got2 = V.bow_vectors(*ora.transform_features(feats, 2), scoring, weighting)
    assert np.array_equal(got2[0][0], bid) and got2[0][1].tobytes() == bval.tobytes()


def test_oracle_transform_vs_product_host_accumulation(tmp_path, oracle):
    path, desc, weight, parent = _vocab_file(tmp_path)
    ora = oracle.VocabOracle(path)
    rng = np.random.RandomState(5)
    feats = rng.randint(0, 256, (300, 32)).astype(np.uint8)
    word, w, node = ora.transform_features(feats, levelsup=2)
    (bid, bval), (fn, fs, ff) = ora.transform(feats, levelsup=2)
    got = V.bow_vectors(word, w, node, 0, 0)
    assert np.array_equal(got[0][0], bid) and got[0][1].tobytes() == bval.tobytes()
    assert np.array_equal(got[1][0], fn) and np.array_equal(got[1][1], fs) and np.array_equal(got[1][2], ff)
    # FeatureVector keys are nodes at depth L - levelsup = 1: children of the root
    assert set(fn.tolist()) <= set(np.nonzero(parent == 0)[0].tolist())
    # L1 score: identical vectors score 1, disjoint vectors 0; product == oracle
    a = got[0]
    assert abs(V.bow_score_l1(a, a) - 1.0) < 1e-12
    b = (a[0] + 100000, a[1])
    assert V.bow_score_l1(a, b) == 0.0
    feats2 = feats.copy()
    feats2[:150] = rng.randint(0, 256, (150, 32))
    c = V.bow_vectors(*ora.transform_features(feats2, 2), 0, 0)[0]
    assert V.bow_score_l1(a, c) == oracle.bow_score_l1(a, c)
    assert 0.0 < V.bow_score_l1(a, c) < 1.0


@pytest.mark.gpu
def test_gpu_bow_transform_matches_oracle(tmp_path, oracle):
    import pilotguru_amd as pg
    path, desc, weight, parent = _vocab_file(tmp_path, k=6, L=4, seed=9)
    ora = oracle.VocabOracle(path)
    voc = V.ORBVocabulary(text_file=path)
    ext = pg.ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240)
    voc.upload(ext)
    rng = np.random.RandomState(11)
    feats = rng.randint(0, 256, (1500, 32)).astype(np.uint8)
    feats[:200] = desc[rng.randint(1, len(desc), 200)]        # exact node descriptors: distance-0 ties
    for levelsup in (4, 2, 0, 7):
        word, w, node = voc.transform_features(feats, levelsup)
        oword, ow, onode = ora.transform_features(feats, levelsup)
        assert np.array_equal(word, oword) and w.tobytes() == ow.tobytes() and np.array_equal(node, onode)
    (bid, bval), fv = voc.transform(feats, levelsup=4)
    (obid, obval), ofv = ora.transform(feats, levelsup=4)
    assert np.array_equal(bid, obid) and bval.tobytes() == obval.tobytes()
    for g, r in zip(fv, ofv):
        assert np.array_equal(g, r)


@pytest.mark.gpu
def test_gpu_vocab_upload_from_device_blob(oracle, tmp_path):
    """The multi-GPU path: the blob arrives in device memory (broadcast receive buffer)."""
    import ctypes as C
    import torch
    import pilotguru_amd as pg
    path, desc, weight, parent = _vocab_file(tmp_path, k=5, L=3, seed=2)
    voc = V.ORBVocabulary(text_file=path)
    ext = pg.ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240)
    t = torch.from_numpy(voc.blob()).cuda()
    ext._check(ext._L.pgorb_vocab_upload_device(ext._h, C.c_void_p(t.data_ptr()), t.numel(),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    voc._ctx = ext
    rng = np.random.RandomState(1)
    feats = rng.randint(0, 256, (100, 32)).astype(np.uint8)
    word, w, node = voc.transform_features(feats, 4)
    oword, ow, onode = oracle.VocabOracle(path).transform_features(feats, 4)
    assert np.array_equal(word, oword) and np.array_equal(node, onode) and w.tobytes() == ow.tobytes()


@pytest.mark.gpu
def test_corrupt_device_blob_is_an_error_code_not_a_fault(tmp_path):
    """a blob that arrives on the device (the broadcast path) is checked structurally by a kernel --
    a child id or child range outside the tree must be refused, and the context must end up without a vocabulary."""
    import ctypes as C
    import torch
    import pilotguru_amd as pg
    from pilotguru_amd.vocab import unpack_vocabulary
    path, desc, weight, parent = _vocab_file(tmp_path, k=4, L=3, seed=3)
    blob = V.ORBVocabulary(text_file=path).blob()
    n = unpack_vocabulary(blob)["nnodes"]
    pad = lambda v: (v + 63) // 64 * 64
    off = 64
    off = pad(off + n * 32); off = pad(off + n * 8); off_parent = off
    off = pad(off + n * 4); off_child0 = off
    off = pad(off + n * 4); off = pad(off + n * 4); off = pad(off + n * 4); off_children = off
    for where, value in ((off_children + 4 * 3, n + 5), (off_child0 + 4 * 2, n), (off_parent + 4 * 5, -2)):
        bad = blob.copy()
        bad[where:where + 4] = np.frombuffer(np.int32(value).tobytes(), np.uint8)
        ext = pg.ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240)
        t = torch.from_numpy(bad).cuda()
        rc = ext._L.pgorb_vocab_upload_device(ext._h, C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == -1, rc                                   # PGORB_E_ARG
        d = torch.zeros((4, 32), dtype=torch.uint8, device="cuda")
        w = torch.zeros(4, dtype=torch.int32, device="cuda"); wt = torch.zeros(4, dtype=torch.float64, device="cuda")
        rc = ext._L.pgorb_bow_transform_device(ext._h, C.c_void_p(d.data_ptr()), 4, 4, C.c_void_p(w.data_ptr()), C.c_void_p(wt.data_ptr()),
                                               C.c_void_p(w.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc != 0                                        # no vocabulary resident


@pytest.mark.gpu
def test_orbvoc_scale_vocabulary(tmp_path, oracle):
    """The size of the real ORBvoc.txt (k = 10, L = 6: 1 111 111 nodes, 10^6 words, 146 MB of text, 66.7 MB
    as a blob; TemplatedVocabulary.h:1337-1420): text -> blob through the product's loader equals the Python
    packer, the tree descent of 2000 descriptors (Frame.cc:399-406: levelsup = 4) equals the oracle's, and the
    stage times are printed (pytest -s)."""
    import time
    import pilotguru_amd as pg
    desc, weight, parent = V.synth_vocabulary_fast(10, 6, seed=11)
    weight[-5000:] = 0.0                                        # some stop words
    path = os.path.join(str(tmp_path), "orbvoc_like.txt")
    t0 = time.perf_counter(); V.write_vocabulary_text_fast(path, 10, 6, desc, weight, parent); t_write = time.perf_counter() - t0
    t0 = time.perf_counter(); voc = V.ORBVocabulary(text_file=path); t_load = time.perf_counter() - t0
    assert (voc.k, voc.L, voc.nnodes, voc.nwords) == (10, 6, 1111111, 1000000)
    assert np.array_equal(voc.blob(), V.pack_vocabulary(10, 6, desc, weight, parent))
    ext = pg.ORBextractor(2000, 1.2, 8, 20, 7, max_width=640, max_height=480)
    t0 = time.perf_counter(); voc.upload(ext); t_up = time.perf_counter() - t0
    rng = np.random.RandomState(4)
    feats = rng.randint(0, 256, (2000, 32)).astype(np.uint8)
    leaves = np.nonzero(np.bincount(parent[1:], minlength=len(parent)) == 0)[0]
    feats[:500] = desc[leaves[rng.randint(0, len(leaves), 500)]]       # exact words: distance-0 paths
    feats[500:800, 3] ^= 0x41
    voc.transform_features(feats, 4)                                    # warm-up
    t0 = time.perf_counter(); word, w, node = voc.transform_features(feats, 4); t_tr = time.perf_counter() - t0
    ora = oracle.VocabOracle(path)
    oword, ow, onode = ora.transform_features(feats, 4)
    assert np.array_equal(word, oword) and w.tobytes() == ow.tobytes() and np.array_equal(node, onode)
    (bid, bval), fv = voc.transform(feats, 4)
    (obid, obval), ofv = ora.transform(feats, 4)
    assert np.array_equal(bid, obid) and bval.tobytes() == obval.tobytes() and all(np.array_equal(g, r) for g, r in zip(fv, ofv))
    print("ORBvoc-scale vocabulary: write text %.1f s, load text -> blob %.2f s (%.1f MB), upload %.3f s, "
          "transform 2000 descriptors (host round trip) %.2f ms" % (t_write, t_load, voc.blob().nbytes / 1e6, t_up, t_tr * 1e3))

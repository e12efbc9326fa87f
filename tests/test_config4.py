"""BASELINE.json configs[3] -- "8 concurrent 1080p rides frame-sharded across 8 x MI355X, RCCL ORB-vocabulary broadcast over
xGMI" -- on the hardware a test box has: ONE MI355X.

Eight extractor contexts (one per ride, `dist.ride_for_rank(r, 8)`, exactly the rides bench.py gives ranks 0..7) live on
the one GPU.  The vocabulary is parsed ONCE from an ORBvoc-sized text file (k = 10, L = 6, 1 111 111 nodes), and reaches
every context through the C ABI's broadcast -- `pgorb_comm_create_local` + `pgorb_vocab_broadcast`, librccl's
ncclCommInitAll / ncclBroadcast (csrc/comm.hip): the group's ranks are the DISTINCT devices (one here), contexts that
share a device receive by a device-to-device copy from their rank's buffer.  Then, for every ride:

  * every frame's keypoints (28-byte records as bytes) and descriptors equal the oracle's;
  * every consecutive match (best-2 Hamming of frame f against f - 1) equals the oracle's;
  * the BoW WORDS, WEIGHTS and NODES of every frame's descriptors (Frame::ComputeBoW, Frame.cc:399-406: levelsup 4)
    equal oracle/bow_oracle.c on the same text file -- SURVEY.md section 7's criterion for the config ("per-GPU BoW words
    equal to rank-0 CPU result"), checked on the values themselves, not on a signature.

Reference: src/optical_trajectories.cc:87-94 (one vocabulary for every System), ORBextractor.cc:1042-1104,
ORBmatcher.cc:1651-1667, TemplatedVocabulary.h:1217-1259."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.gpu

W, H, NF, NRIDES, NFRAMES = 1920, 1080, 2000, 8, 8


@pytest.fixture(scope="module")
def orbvoc_like(tmp_path_factory):
    from pilotguru_amd import vocab as V
    d = tmp_path_factory.mktemp("voc")
    desc, weight, parent = V.synth_vocabulary_fast(10, 6, seed=5)
    weight[-3000:] = 0.0                                         # stop words: weight 0 is dropped from the BowVector
    path = os.path.join(str(d), "orbvoc_like.txt")
    V.write_vocabulary_text_fast(path, 10, 6, desc, weight, parent)
    return path


def test_eight_rides_on_eight_contexts_with_one_broadcast_vocabulary(orbvoc_like, oracle):
    import torch
    import pilotguru_amd as pg
    from pilotguru_amd import dist as pgd
    from pilotguru_amd import vocab as V
    from pilotguru_amd.synth import synth_ride
    from _oracle_pool import oracle_ride

    rides = [pgd.ride_for_rank(r, NRIDES)[0] for r in range(NRIDES)]
    assert rides == list(range(NRIDES))                          # ride r -> rank r: config 4's sharding
    exts = [pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=NFRAMES) for _ in rides]

    # ---- the one collective: text parsed once, one broadcast from context 3 (any root must do) ----
    voc = V.ORBVocabulary(text_file=orbvoc_like)
    assert (voc.k, voc.L, voc.nnodes) == (10, 6, 1111111)
    comm = pgd.VocabularyComm.local(exts)
    assert comm.ranks() == 1                                     # eight contexts, one device: one RCCL rank
    seconds = comm.broadcast(voc, root=3)
    assert seconds > 0
    comm.close()
    ora_voc = oracle.VocabOracle(orbvoc_like)

    cap = exts[0].max_keypoints(W, H)
    streams = [torch.cuda.Stream() for _ in rides]
    frames_host = [synth_ride(r, W, H, NFRAMES) for r in rides]
    out = []
    # eight rides concurrently: every context on its own HIP stream, nothing synchronised until all are queued
    for ext, ride, s in zip(exts, frames_host, streams):
        with torch.cuda.stream(s):
            fr = torch.from_numpy(ride).cuda(non_blocking=False)
            kps, desc, n = ext.extract_batch_device(fr)
            pq = torch.arange(1, NFRAMES, dtype=torch.int32, device="cuda")
            pt = torch.arange(0, NFRAMES - 1, dtype=torch.int32, device="cuda")
            m = ext.match_batch_device(desc, n, pq, pt)
            out.append((fr, kps, desc, n, m))
    torch.cuda.synchronize()
    for ext in exts:
        ext.check_async()

    for r, (ext, ride, (fr, kps, desc, n, (bi, b1, b2))) in enumerate(zip(exts, frames_host, out)):
        oext, omatch = oracle_ride(list(ride), (NF, 1.2, 8, 20, 7))
        nh = n.cpu().numpy()
        assert nh.min() >= NF - 1, "ride %d: a level missed its quota (scene acceptance, SURVEY.md 8d)" % r
        bi_h, b1_h, b2_h = bi.cpu().numpy(), b1.cpu().numpy().view(np.uint16), b2.cpu().numpy().view(np.uint16)
        for f in range(NFRAMES):
            okp, odesc = oext[f]
            assert nh[f] * 28 == len(okp), "ride %d frame %d: %d keypoints, oracle %d" % (r, f, nh[f], len(okp) // 28)
            assert kps[f, :nh[f]].cpu().numpy().tobytes() == okp, "ride %d: keypoints of frame %d" % (r, f)
            d = desc[f, :nh[f]].cpu().numpy()
            assert d.tobytes() == odesc, "ride %d: descriptors of frame %d" % (r, f)
            if f >= 1:
                obi, ob1, ob2 = omatch[f - 1]
                assert bi_h[f - 1, :nh[f]].tobytes() == obi and b1_h[f - 1, :nh[f]].tobytes() == ob1 and \
                    b2_h[f - 1, :nh[f]].tobytes() == ob2, "ride %d: best-2 match of frame %d vs %d" % (r, f, f - 1)
            # Frame::ComputeBoW on THIS ride's context, with the vocabulary that arrived through the broadcast
            voc._ctx = ext
            word, weight, node = voc.transform_features(d, 4)
            oword, oweight, onode = ora_voc.transform_features(d, 4)
            assert np.array_equal(word, oword), "ride %d frame %d: BoW words" % (r, f)
            assert weight.tobytes() == oweight.tobytes(), "ride %d frame %d: BoW weights" % (r, f)
            assert np.array_equal(node, onode), "ride %d frame %d: BoW nodes" % (r, f)
            if f == 0:
                (bid, bval), fv = voc.transform(d, 4)
                (obid, obval), ofv = ora_voc.transform(d, 4)
                assert np.array_equal(bid, obid) and bval.tobytes() == obval.tobytes()
                assert all(np.array_equal(g, o) for g, o in zip(fv, ofv))


def test_broadcast_group_errors_are_codes_not_crashes():
    """The group API's argument checks (include/pgorb.h): a root outside the group, a missing vocabulary on the root."""
    import pilotguru_amd as pg
    from pilotguru_amd import dist as pgd
    from pilotguru_amd import vocab as V
    from pilotguru_amd._lib import PgorbError
    exts = [pg.ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240) for _ in range(2)]
    comm = pgd.VocabularyComm.local(exts)
    voc = V.ORBVocabulary(blob=V.synth_vocabulary_blob(4, 3, seed=2))
    with pytest.raises(PgorbError):
        comm.broadcast(voc, root=2)
    with pytest.raises(PgorbError):
        comm.broadcast(None, root=0)
    comm.broadcast(voc, root=1)
    rng = np.random.RandomState(0)
    feats = rng.randint(0, 256, (300, 32)).astype(np.uint8)
    res = []
    for e in exts:
        voc._ctx = e
        res.append(voc.transform_features(feats, 2))
    assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))


def test_rank_form_of_the_broadcast_as_a_one_rank_group():
    """pgorb_comm_unique_id + pgorb_comm_create_rank (ncclCommInitRank) with nranks = 1: the per-process form bench.py's
    ranks use, as far as one GPU can take it."""
    import ctypes as C
    import pilotguru_amd as pg
    from pilotguru_amd import _lib, vocab as V
    L = _lib.lib()
    ext = pg.ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240)
    ident = (C.c_uint8 * 128)()
    assert L.pgorb_comm_unique_id(ident) == 0 and any(ident)
    h = C.c_void_p()
    ext._check(L.pgorb_comm_create_rank(ext._h, 0, 1, ident, C.byref(h)))
    voc = V.ORBVocabulary(blob=V.synth_vocabulary_blob(5, 4, seed=3))
    sec = C.c_double()
    ext._check(L.pgorb_vocab_broadcast(h, 0, voc._h, C.byref(sec)))
    L.pgorb_comm_destroy(h)
    voc._ctx = ext
    feats = np.random.RandomState(1).randint(0, 256, (200, 32)).astype(np.uint8)
    word, weight, node = voc.transform_features(feats, 2)
    assert len(word) == 200 and word.max() < voc.nwords

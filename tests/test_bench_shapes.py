"""Parity at the shapes bench.py measures, and the N > 1 code on one GPU.

 * every frame of a ride at BASELINE.json configs[1] (1920x1080 / 2000, 32 frames) and configs[2]
   (3840x2160 / 4000, 8 frames): keypoints (28-byte records as bytes), descriptors AND the best-2
   Hamming match of every frame against its predecessor, against the oracle (fanned over a process
   pool) -- SURVEY.md section 7's config-2 criterion "bit-exact on every frame of the ride";
 * the same for the driving-like scene class (sky / asphalt: the minThFAST retry is live);
 * configs[3]'s code path (RCCL process group, vocabulary broadcast, device-side blob upload, the
   cross-rank BoW signature) as a one-rank torchrun of bench.py on the one GPU a test box has;
 * one ride split over two contexts with dist.frame_chunk_for_rank reproduces the unsplit ride.

Reference semantics: ORBextractor.cc:1042-1104 (operator()), ORBmatcher.cc:1651-1667
(DescriptorDistance), src/optical_trajectories.cc:87-94 (one vocabulary for every System)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pilotguru_amd.synth import synth_ride, synth_ride_road

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

pytestmark = pytest.mark.gpu


def _ride_vs_oracle(ride, nf, batch):
    import torch
    import pilotguru_amd as pg
    from _oracle_pool import oracle_ride
    B, h, w = ride.shape
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch)
    cap = ext.max_keypoints(w, h)
    oext, omatch = oracle_ride(list(ride), (nf, 1.2, 8, 20, 7))
    kall, dall, nall = [], [], []
    for b0 in range(0, B, batch):                                   # the bench's call, batch by batch
        fr = torch.from_numpy(ride[b0:b0 + batch]).cuda()
        kps, desc, n = ext.extract_batch_device(fr)
        ext.check_async()
        kall.append(kps.clone()); dall.append(desc.clone()); nall.append(n.clone())
    kps, desc, n = torch.cat(kall), torch.cat(dall), torch.cat(nall)
    nh = n.cpu().numpy()
    for f in range(B):
        okp, odesc = oext[f]
        assert nh[f] * 28 == len(okp), "frame %d: %d keypoints, oracle %d" % (f, nh[f], len(okp) // 28)
        assert kps[f, :nh[f]].cpu().numpy().tobytes() == okp, "keypoints of frame %d" % f
        assert desc[f, :nh[f]].cpu().numpy().tobytes() == odesc, "descriptors of frame %d" % f
    pq = torch.arange(1, B, dtype=torch.int32, device="cuda")
    pt = torch.arange(0, B - 1, dtype=torch.int32, device="cuda")
    bi, b1, b2 = ext.match_batch_device(desc, n, pq, pt)
    ext.check_async()
    torch.cuda.synchronize()
    bi, b1, b2 = bi.cpu().numpy(), b1.cpu().numpy().view(np.uint16), b2.cpu().numpy().view(np.uint16)
    for f in range(1, B):
        obi, ob1, ob2 = omatch[f - 1]
        assert bi[f - 1, :nh[f]].tobytes() == obi and b1[f - 1, :nh[f]].tobytes() == ob1 and \
            b2[f - 1, :nh[f]].tobytes() == ob2, "best-2 match of frame %d vs %d" % (f, f - 1)
    return nh


@pytest.mark.parametrize("w,h,nf,B,batch", [
    (1920, 1080, 2000, 32, 16), (3840, 2160, 4000, 8, 4),
    (1920, 1080, 2000, 128, 128)])      # the bench's own plan: batch 128 (the XCD-balanced cell order and K4-6's tile order depend on it)
def test_every_frame_of_a_ride_at_bench_shapes(w, h, nf, B, batch):
    nh = _ride_vs_oracle(synth_ride(0, w, h, B), nf, batch)
    assert nh.min() >= nf - 1                      # acceptance of the scene: every level fills its quota


@pytest.mark.parametrize("w,h,nf,B", [(1920, 1080, 2000, 12), (640, 480, 1000, 6)])
def test_driving_scene_with_flat_regions(w, h, nf, B, oracle):
    """>= 40 % of the cells are sky / asphalt: no corner at iniThFAST, the per-cell minThFAST retry
    (ORBextractor.cc:812-816) decides what those cells contribute."""
    ride = synth_ride_road(3, w, h, B)
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    ora.extract(ride[0])
    resp = ora.level_candidates(0)["response"]
    assert (resp < 20).mean() > 0.4                # the retry produced a large share of level 0's candidates
    _ride_vs_oracle(ride, nf, B)


def test_single_ride_split_over_two_contexts():
    """SURVEY.md section 8(e), single ride over several GPUs: contiguous chunks with a one-frame overlap,
    nothing exchanged.  Two contexts (standing in for two ranks) each take their chunk; owned frames
    and owned (f, f-1) matches together equal the unsplit run."""
    import torch
    import pilotguru_amd as pg
    from pilotguru_amd import dist as pgd
    w, h, nf, B = 640, 480, 1000, 9
    ride = synth_ride(11, w, h, B)

    def run(frames):
        ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=len(frames))
        kps, desc, n = ext.extract_batch_device(torch.from_numpy(frames).cuda())
        m = len(frames) - 1
        pq = torch.arange(1, m + 1, dtype=torch.int32, device="cuda")
        pt = torch.arange(0, m, dtype=torch.int32, device="cuda")
        bi, b1, b2 = ext.match_batch_device(desc, n, pq, pt) if m else (None, None, None)
        ext.check_async()
        torch.cuda.synchronize()
        nh = n.cpu().numpy()
        out = []
        for f in range(len(frames)):
            rec = [kps[f, :nh[f]].cpu().numpy().tobytes(), desc[f, :nh[f]].cpu().numpy().tobytes(), None]
            if f >= 1:
                rec[2] = (bi[f - 1, :nh[f]].cpu().numpy().tobytes(), b1[f - 1, :nh[f]].cpu().numpy().tobytes(),
                          b2[f - 1, :nh[f]].cpu().numpy().tobytes())
            out.append(rec)
        return out

    whole = run(ride)
    for world in (2, 3):
        owned = {}
        for rank in range(world):
            fe, fo, stop = pgd.frame_chunk_for_rank(B, rank, world)
            part = run(ride[fe:stop])
            for f in range(fo, stop):
                owned[f] = part[f - fe]
        assert sorted(owned) == list(range(B))
        for f in range(B):
            assert owned[f][0] == whole[f][0] and owned[f][1] == whole[f][1]
            if f >= 1:
                assert owned[f][2] == whole[f][2], "match of frame %d lost at a chunk border" % f


def test_multi_gpu_code_path_on_one_gpu():
    """bench.py's N > 1 branch as a one-rank job under torch.distributed.run: init_process_group("nccl") (RCCL) as the
    control plane, the ORBvoc-sized vocabulary parsed from text on rank 0 and broadcast through the C ABI
    (pgorb_comm_unique_id / pgorb_comm_create_rank / pgorb_vocab_broadcast: librccl called directly), the BoW transform
    of the rank's own first frame gathered to rank 0 and compared with the oracle's words / weights / nodes, the barriers
    and the max-over-ranks timing (closing barrier outside the interval `value` is built on)."""
    env = dict(os.environ, PGORB_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["config"]["vocab_broadcast_bytes"] > 60 << 20         # the ORBvoc-sized blob (k=10, L=6) went through RCCL
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["verified"] is True
    assert out["config"]["bow_words_equal_oracle_on_every_rank"] is True and out["config"]["vocab_broadcast_s"] > 0
    assert out["barrier_inclusive_seconds"] >= out["max_rank_seconds"]
    # ONE librccl in the process: the library took the copy torch.distributed had already mapped (csrc/comm.hip, RTLD_NOLOAD first)
    assert out["config"]["vocab_broadcast"].startswith("pgorb_vocab_broadcast")
    info = out["config"]["vocab_broadcast_librccl"]
    assert len(info["mapped"]) == 1 and os.path.realpath(info["path"]) == os.path.realpath(info["mapped"][0]), info
    assert info["held_by_process_before"] is True


@pytest.mark.parametrize("w,h,nf,total,batch,depth", [(640, 480, 1000, 23, 8, 3), (1920, 1080, 2000, 20, 8, 2)])
def test_streamed_ingest_bit_exact(w, h, nf, total, batch, depth):
    """Frames that start in host memory: page-locked slots, three HIP streams (upload / kernels / download),
    several batches in flight, a ragged last batch -- every frame's keypoints and descriptors and every frame's
    best-2 match against its predecessor ACROSS batch borders against the oracle
    (src/slam/track_image_sequence.cc:43-47: frame after frame of one ride)."""
    import pilotguru_amd as pg
    from _oracle_pool import oracle_ride
    ride = synth_ride(21, w, h, total)
    oext, omatch = oracle_ride(list(ride), (nf, 1.2, 8, 20, 7))
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch)
    st = pg.FrameStream(ext, w, h, batch, depth)
    chunks = [(b0, min(batch, total - b0)) for b0 in range(0, total, batch)]
    results = {}
    inflight = []
    for i, (b0, nb) in enumerate(chunks):
        slot = i % depth
        if len(inflight) == depth:                       # the slot about to be reused must be collected first
            j, s0 = inflight.pop(0)
            results[j] = [np.array(a) for a in st.wait(s0)]
        st.input(slot)[:nb] = ride[b0:b0 + nb]           # "the decoder" writes into page-locked memory
        st.submit(slot, nb)
        inflight.append((i, slot))
    for j, s0 in inflight:
        results[j] = [np.array(a) for a in st.wait(s0)]
    st.close()
    for i, (b0, nb) in enumerate(chunks):
        n, kps, desc, bi, b1, b2 = results[i]
        assert len(n) == nb
        for k in range(nb):
            f = b0 + k
            okp, odesc = oext[f]
            assert n[k] * 28 == len(okp) and kps[k, :n[k]].tobytes() == okp and desc[k, :n[k]].tobytes() == odesc, "frame %d" % f
            if f == 0:
                assert np.all(bi[k, :n[k]] == -1)        # no predecessor
            else:
                obi, ob1, ob2 = omatch[f - 1]
                assert bi[k, :n[k]].tobytes() == obi and b1[k, :n[k]].tobytes() == ob1 and b2[k, :n[k]].tobytes() == ob2, \
                    "match of frame %d vs %d" % (f, f - 1)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,nf,total,batch,depth,cn,rgb,rot,vf,hf", [
    (640, 480, 1000, 9, 4, 3, 3, True, 90, False, True),       # phone video: RGB24, rotated by the metadata, --horizontal_flip
    (1280, 720, 1500, 5, 2, 2, 3, False, 180, True, False),    # BGR24, upside down, --vertical_flip
    (640, 480, 1000, 5, 3, 2, 4, True, 270, False, False),     # RGBA
    (1920, 1080, 2000, 3, 2, 2, 3, True, 0, False, False),     # the bench's RGB24 leg
    (640, 480, 1000, 5, 3, 2, 1, True, 90, True, True)])       # grey, geometry only
def test_streamed_rgb_ride_with_the_readers_geometry(oracle, w, h, nf, total, batch, depth, cn, rgb, rot, vf, hf):
    """pgorb_stream_create_ingest: the slots hold the frames EXACTLY as the reference's reader decodes them (RGB24,
    src/io/image_sequence_reader.cc:138-208), rotation by the metadata (:186-205), the wrapper's flips (:53-58) and
    Tracking's cvtColor (Tracking.cc:247-260) run on the device in front of the pyramid; every frame and every match
    across batch borders against the oracle run on the oracle-ingested grey frames."""
    import pilotguru_amd as pg
    from _oracle_pool import oracle_ride
    ride = synth_ride(33, w, h, total)
    if cn == 1:
        src = ride
    else:
        planes = [ride, np.roll(ride, 3, axis=2), 255 - ride] + ([np.full_like(ride, 255)] if cn == 4 else [])
        src = np.ascontiguousarray(np.stack(planes, axis=3))
    upright = []
    for f in range(total):
        up = oracle.ingest_geometry(src[f], rot, vf, hf)
        if cn > 1:
            up = oracle.rgb_to_gray(np.ascontiguousarray(up[:, :, :3] if rgb else up[:, :, 2::-1]))
        upright.append(np.ascontiguousarray(up))
    ow, oh = (h, w) if rot in (90, 270) else (w, h)
    assert upright[0].shape == (oh, ow)
    oext, omatch = oracle_ride(upright, (nf, 1.2, 8, 20, 7))
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=ow, max_height=oh, max_batch=batch)
    st = pg.FrameStream(ext, w, h, batch, depth, channels=cn, rgb_order=rgb, rotate_degrees=rot, vertical_flip=vf, horizontal_flip=hf)
    assert st.input(0).shape == ((batch, h, w) if cn == 1 else (batch, h, w, cn))
    chunks = [(b0, min(batch, total - b0)) for b0 in range(0, total, batch)]
    results, inflight = {}, []
    for i, (b0, nb) in enumerate(chunks):
        slot = i % depth
        if len(inflight) == depth:
            j, s0 = inflight.pop(0)
            results[j] = [np.array(a) for a in st.wait(s0)]
        st.input(slot)[:nb] = src[b0:b0 + nb]
        st.submit(slot, nb)
        inflight.append((i, slot))
    for j, s0 in inflight:
        results[j] = [np.array(a) for a in st.wait(s0)]
    st.close()
    for i, (b0, nb) in enumerate(chunks):
        n, kps, desc, bi, b1, b2 = results[i]
        for k in range(nb):
            f = b0 + k
            okp, odesc = oext[f]
            assert n[k] * 28 == len(okp) and kps[k, :n[k]].tobytes() == okp and desc[k, :n[k]].tobytes() == odesc, "frame %d" % f
            if f:
                obi, ob1, ob2 = omatch[f - 1]
                assert bi[k, :n[k]].tobytes() == obi and b1[k, :n[k]].tobytes() == ob1 and b2[k, :n[k]].tobytes() == ob2, "match %d" % f


@pytest.mark.gpu
def test_destroying_the_context_takes_its_streams_along():
    """pgorb_destroy with a live stream used to leave the stream pointing at a freed context."""
    import ctypes as C
    import pilotguru_amd as pg
    ext = pg.ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240, max_batch=2)
    st = pg.FrameStream(ext, 320, 240, 2, 2)
    st.input(0)[:] = synth_ride(1, 320, 240, 2)
    st.submit(0)
    L, h = ext._L, ext._h
    ext._h = None                                  # the Python object must not destroy it a second time
    st._s = None
    L.pgorb_destroy(h)                             # waits for the batch in flight, frees the stream, then the context


@pytest.mark.parametrize("w,h,nf,total,batch,depth,bow", [
    (640, 480, 1000, 11, 4, 3, True), (1280, 720, 1500, 7, 3, 2, False),
    # the INITIALISATION workload: until the map is initialised the reference extracts with 2 * nFeatures
    # (Tracking.cc:137-143, :262-264) and runs SearchForInitialization(window 100, ratio 0.9) on those frames (:596-597)
    (1920, 1080, 4000, 5, 2, 2, False)])
def test_streamed_front_end_stage_against_the_oracle(w, h, nf, total, batch, depth, bow, tmp_path):
    """pgorb_stream_frontend: what the tracking thread does with every fresh Frame, on the device per batch -- the
    SearchForInitialization of (previous frame, frame) as MonocularInitialization calls it (Tracking.cc:583-597,
    ORBmatcher.cc:407-522) ACROSS batch borders, and ORBVocabulary::transform of every descriptor (Frame.cc:399-406) --
    against the oracle, frame by frame."""
    import pilotguru_amd as pg
    from pilotguru_amd import vocab as V
    from oracle import orb_oracle
    from _oracle_pool import oracle_ride
    ride = synth_ride(5, w, h, total, dx=3, dy=1)
    oext, _ = oracle_ride(list(ride), (nf, 1.2, 8, 20, 7), match=False)
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch)
    voc = None
    if bow:
        desc, weight, parent = V.synth_vocabulary(6, 4, seed=2)
        path = os.path.join(str(tmp_path), "voc.txt")
        V.write_vocabulary_text(path, 6, 4, desc, weight, parent)
        V.ORBVocabulary(text_file=path).upload(ext)
        voc = orb_oracle.VocabOracle(path)
    st = pg.FrameStream(ext, w, h, batch, depth)
    bounds = (0.0, float(w), 0.0, float(h))
    st.frontend(bounds, 100, 0.9, True, 4 if bow else -1)
    chunks = [(b0, min(batch, total - b0)) for b0 in range(0, total, batch)]
    res, inflight = {}, []

    def collect(j, s0):
        out = [np.array(a) for a in st.wait(s0)]
        nb, cap = out[1].shape
        res[j] = out + [None if a is None else np.array(a) for a in st.frontend_results(s0, nb, cap)]
    for i, (b0, nb) in enumerate(chunks):
        slot = i % depth
        if len(inflight) == depth:
            collect(*inflight.pop(0))
        st.input(slot)[:nb] = ride[b0:b0 + nb]
        st.submit(slot, nb)
        inflight.append((i, slot))
    for j, s0 in inflight:
        collect(j, s0)
    st.close()
    prev = None
    for i, (b0, nb) in enumerate(chunks):
        n, kps, desc, _, _, _, m12, nm, word, wt, node = res[i]
        for k in range(nb):
            f = b0 + k
            okp, odesc = oext[f]
            assert kps[k, :n[k]].tobytes() == okp and desc[k, :n[k]].tobytes() == odesc, "frame %d" % f
            kp = np.frombuffer(okp, orb_oracle.KEYPOINT_DTYPE)
            de = np.frombuffer(odesc, np.uint8).reshape(-1, 32)
            if prev is None:
                assert nm[k] == 0
            else:
                pk, pd = prev
                onm, om12, _ = orb_oracle.search_for_initialization(pk, pd, kp, de, bounds, np.stack([pk["x"], pk["y"]], 1).astype(np.float32),
                                                                    100, 0.9, True)
                assert nm[k] == onm and onm > 30, "SearchForInitialization count of frame %d" % f
                assert np.array_equal(m12[k, :len(pk)], om12), "vnMatches12 of frame %d" % f
            if bow:
                ow, owt, on = voc.transform_features(de, 4)
                assert np.array_equal(word[k, :n[k]], ow) and np.array_equal(wt[k, :n[k]], owt) and np.array_equal(node[k, :n[k]], on)
            prev = (kp, de)


def test_stream_error_paths():
    """The stream refuses what it cannot do, with a message: results of a slot that holds no batch, a second submit of an
    uncollected slot, the front-end stage with a BoW transform but no vocabulary, front-end results when the stage is off."""
    import pilotguru_amd as pg
    from pilotguru_amd import _lib
    w, h, nf, batch = 320, 240, 500, 2
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch)
    st = pg.FrameStream(ext, w, h, batch, 2)
    with pytest.raises(_lib.PgorbError, match="no batch in flight"):
        st.wait(0)
    with pytest.raises(_lib.PgorbError, match="not enabled"):
        st.frontend_results(0, batch, ext.max_keypoints(w, h))
    with pytest.raises(_lib.PgorbError):                       # BoW asked for, no vocabulary resident
        st.frontend((0.0, float(w), 0.0, float(h)), 100, 0.9, True, 4)
    with pytest.raises(_lib.PgorbError):
        st.frontend((0.0, 0.0, 0.0, float(h)), 100, 0.9, True, -1)      # empty image bounds
    st.input(0)[:] = synth_ride(2, w, h, batch)
    st.submit(0)
    with pytest.raises(_lib.PgorbError, match="not collected"):
        st.submit(0)
    with pytest.raises(_lib.PgorbError, match="in flight"):
        st.frontend((0.0, float(w), 0.0, float(h)), 100, 0.9, True, -1)
    n = st.wait(0)[0]
    assert len(n) == batch and n.min() > 100
    with pytest.raises(Exception):
        pg.FrameStream(ext, w, h, batch + 1, 2)                 # more than max_batch
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,nf,total,batch,depth,lanes", [
    (640, 480, 1000, 29, 8, 3, 2), (640, 480, 1000, 13, 4, 4, 3), (1920, 1080, 2000, 20, 8, 2, 2), (640, 480, 1000, 9, 4, 2, 1)])
def test_device_resident_stream_with_batches_in_flight(w, h, nf, total, batch, depth, lanes):
    """pgorb_stream_create_device / _submit_device / _wait_device (round 5): resident frames, `lanes` batches in flight
    inside the library on sibling working sets, results on the device.  Every frame's keypoints and descriptors and every
    frame's best-2 match against its predecessor -- ACROSS the border between two batches that ran concurrently on
    different lanes -- against the oracle; a ragged last batch; slots reused (more batches than slots)."""
    import torch
    import pilotguru_amd as pg
    from _oracle_pool import oracle_ride
    ride = synth_ride(41, w, h, total)
    oext, omatch = oracle_ride(list(ride), (nf, 1.2, 8, 20, 7))
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch)
    st = pg.DeviceFrameStream(ext, w, h, batch, depth, lanes)
    assert st.lanes() == lanes
    frames = torch.from_numpy(ride).cuda()
    chunks = [(b0, min(batch, total - b0)) for b0 in range(0, total, batch)]
    results, inflight = {}, []

    def collect(j, slot):
        results[j] = [t.cpu().numpy() for t in st.wait(slot)]
    for i, (b0, nb) in enumerate(chunks):
        slot = i % depth
        if len(inflight) == depth:
            collect(*inflight.pop(0))
        st.submit(slot, frames[b0:b0 + nb])
        inflight.append((i, slot))
    for j, slot in inflight:
        collect(j, slot)
    st.close()
    for i, (b0, nb) in enumerate(chunks):
        n, kps, desc, bi, b1, b2 = results[i]
        assert len(n) == nb
        for k in range(nb):
            f = b0 + k
            okp, odesc = oext[f]
            assert n[k] * 28 == len(okp) and kps[k, :n[k]].tobytes() == okp and desc[k, :n[k]].tobytes() == odesc, "frame %d" % f
            if f == 0:
                assert np.all(bi[k, :n[k]] == -1)
            else:
                obi, ob1, ob2 = omatch[f - 1]
                assert bi[k, :n[k]].tobytes() == obi and b1[k, :n[k]].view(np.uint16).tobytes() == ob1 and \
                    b2[k, :n[k]].view(np.uint16).tobytes() == ob2, "match of frame %d vs %d" % (f, f - 1)


@pytest.mark.gpu
def test_device_resident_stream_front_end_stage_equals_the_host_stream():
    """The front-end stage (grid, SearchForInitialization of every frame against its predecessor, BoW transform) behind a
    device-resident stream with two lanes equals the host-frame stream's (which tests/test_frame_matcher.py and
    test_streamed_front_end_stage_against_the_oracle hold to the oracle): the stage's cross-batch state -- the previous
    frame's keypoints, vbPrevMatched -- is chained in submission order although the batches extract concurrently."""
    import torch
    import pilotguru_amd as pg
    from pilotguru_amd import vocab as V
    w, h, nf, total, batch = 640, 480, 1000, 19, 6
    ride = synth_ride(5, w, h, total, dx=3, dy=1)
    chunks = [(b0, min(batch, total - b0)) for b0 in range(0, total, batch)]

    def run(device_form):
        ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch)
        V.ORBVocabulary(blob=V.synth_vocabulary_blob(8, 4, seed=3)).upload(ext)
        st = pg.DeviceFrameStream(ext, w, h, batch, 2, 2) if device_form else pg.FrameStream(ext, w, h, batch, 2)
        st.frontend((0.0, float(w), 0.0, float(h)), 100, 0.9, True, 3)
        frames = torch.from_numpy(ride).cuda() if device_form else None
        out = []
        pending = []

        def collect(slot, nb):
            res = st.wait(slot)
            cap = res[1].shape[1]
            if device_form:
                ptr = [pg.orb.C.c_void_p() for _ in range(5)]
                ext._check(ext._L.pgorb_stream_frontend_results(st._s, slot, *[pg.orb.C.byref(p) for p in ptr]))
                dev = torch.device("cuda", 0)
                m12 = pg.orb._device_tensor(ptr[0].value, nb * cap * 4, dev).view(torch.int32).reshape(nb, cap).cpu().numpy()
                nm = pg.orb._device_tensor(ptr[1].value, nb * 4, dev).view(torch.int32).cpu().numpy()
                word = pg.orb._device_tensor(ptr[2].value, nb * cap * 4, dev).view(torch.int32).reshape(nb, cap).cpu().numpy().view(np.uint32)
                n = res[0].cpu().numpy()
            else:
                m12, nm, word, _, _ = st.frontend_results(slot, nb, cap)
                n = np.array(res[0])
            for k in range(nb):
                # matches12 of the pair (previous frame, frame k) is indexed by the PREVIOUS frame's keypoints
                nprev = out[-1][0] if out else 0
                out.append((int(n[k]), int(nm[k]), m12[k, :nprev].copy() if out else None, word[k, :n[k]].copy()))
        for i, (b0, nb) in enumerate(chunks):
            slot = i % 2
            if len(pending) == 2:
                collect(*pending.pop(0))
            if device_form:
                st.submit(slot, frames[b0:b0 + nb])
            else:
                st.input(slot)[:nb] = ride[b0:b0 + nb]
                st.submit(slot, nb)
            pending.append((slot, nb))
        for p in pending:
            collect(*p)
        st.close()
        return out
    host, dev = run(False), run(True)
    assert len(host) == len(dev) == total
    for f, (a, b) in enumerate(zip(host, dev)):
        assert a[0] == b[0] and np.array_equal(a[3], b[3]), "frame %d: keypoint count / BoW words" % f
        if f:
            assert a[1] == b[1] and np.array_equal(a[2], b[2]), "frame %d: SearchForInitialization against its predecessor" % f
    assert sum(a[1] for a in host[1:]) > 100

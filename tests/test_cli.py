"""The optical_trajectories-compatible CLI (pilotguru_amd/host/optical_trajectories, C++ over the
C ABI): flag surface and CHECKs of src/optical_trajectories.cc:36-79, the trajectory JSON
writer (src/io/json_converters.cc:6-96 + nlohmann dump(2)) against a committed fixture, and
-- on the GPU -- the front-end run against the Python path."""
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(ROOT, "pilotguru_amd", "host", "optical_trajectories")


def _cli(*args):
    return subprocess.run([CLI] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)


def test_cli_checks_required_flags():
    assert os.path.exists(CLI), "build it: make -C pilotguru_amd/csrc"
    r = _cli()
    assert r.returncode != 0 and "!FLAGS_vocabulary_file.empty()" in r.stderr          # :77
    r = _cli("--vocabulary_file=v.txt")
    assert r.returncode != 0 and "!FLAGS_camera_settings.empty()" in r.stderr          # :78
    r = _cli("--vocabulary_file=v.txt", "--camera_settings", "c.yml")
    assert r.returncode != 0 and "!FLAGS_in_video.empty()" in r.stderr                 # :79
    r = _cli("--no_such_flag=1")
    assert r.returncode != 0 and "unknown command line flag" in r.stderr
    # boolean forms of gflags are accepted up to the first failing CHECK
    r = _cli("--novisualize", "--vertical_flip", "--horizontal_flip=false", "--output_per_segment_videos",
             "--rotation_smooth_sigma=5", "--out_dir=/tmp")
    assert "unknown command line flag" not in r.stderr


def test_trajectory_json_writer_matches_fixture(tmp_path):
    r = _cli("--trajectory_in=" + os.path.join(HERE, "golden", "trajectory_in.txt"), "--out_dir=" + str(tmp_path))
    assert r.returncode == 0, r.stderr
    got = open(os.path.join(str(tmp_path), "trajectory-0.json")).read()
    want = open(os.path.join(HERE, "golden", "trajectory_expected.json")).read()
    assert got == want
    d = json.loads(got)                                               # schema of json_converters.cc:56-96
    assert sorted(d) == ["plane", "trajectory"]
    p = d["trajectory"][1]
    assert sorted(p) == ["angular_velocity", "frame_id", "is_lost", "planar_direction", "pose", "time_usec"]
    assert sorted(p["pose"]) == ["rotation", "translation"] and sorted(p["pose"]["rotation"]) == ["w", "x", "y", "z"]
    assert d["trajectory"][0]["angular_velocity"] == 0 and '"angular_velocity": 0,' in got   # integer 0 (:83)
    assert abs(p["angular_velocity"] - 0.01 / (0.033333 + 1e-10)) < 1e-12


def test_trajectory_json_layout_against_a_real_nlohmann_dump(tmp_path):
    """The writer's STRUCTURE (key order, indentation, one array element per line, integer 0, null) against a file a real
    nlohmann::json produced: tests/golden/make_trajectory_layout.cc assembles the object as src/io/json_converters.cc:6-96
    does and dumps it with the 3.1.1 header of the build image; the values are ones 2.1.1 (the reference's version) and
    3.x print identically.  When the header is present (this container) the generator is re-run and must reproduce the fixture."""
    import shutil
    src = os.path.join(HERE, "golden", "trajectory_layout_in.txt")
    want = open(os.path.join(HERE, "golden", "trajectory_layout_expected.json")).read()
    r = _cli("--trajectory_in=" + src, "--out_dir=" + str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert open(os.path.join(str(tmp_path), "trajectory-0.json")).read() == want
    assert '"x": null' not in want and "null" in want and '"angular_velocity": 0,' in want
    if os.path.exists("/opt/conda/include/json.hpp") and shutil.which("g++"):
        exe = os.path.join(str(tmp_path), "mtl")
        subprocess.run(["g++", "-std=c++11", "-I/opt/conda/include", os.path.join(HERE, "golden", "make_trajectory_layout.cc"), "-o", exe], check=True)
        got = subprocess.run([exe, src], stdout=subprocess.PIPE, universal_newlines=True, check=True).stdout
        assert got == want


def test_trajectory_json_non_finite_numbers_become_null(tmp_path):
    """nlohmann 2.1.1 stores a non-finite float as null, so the reference's files stay valid JSON."""
    src = open(os.path.join(HERE, "golden", "trajectory_in.txt")).read().split("\n")
    idx = [i for i, l in enumerate(src) if l.startswith("33333 ")][0]
    f = src[idx].split()
    f[3], f[12] = "nan", "inf"                                       # tx of point 1, its turn angle
    src[idx] = " ".join(f)
    path = os.path.join(str(tmp_path), "traj_nan.txt")
    open(path, "w").write("\n".join(src))
    r = _cli("--trajectory_in=" + path, "--out_dir=" + str(tmp_path))
    assert r.returncode == 0, r.stderr
    got = open(os.path.join(str(tmp_path), "trajectory-0.json")).read()
    d = json.loads(got)                                               # parses: no bare nan / inf tokens
    assert d["trajectory"][1]["pose"]["translation"][0] is None
    assert d["trajectory"][1]["angular_velocity"] is None
    assert "nan" not in got and "inf" not in got


def test_cli_rejects_hostile_frame_patterns(tmp_path):
    """--in_video is never used as a printf format: one %d / %0Nd conversion or nothing."""
    settings = os.path.join(str(tmp_path), "cam.yml")
    open(settings, "w").write("%YAML:1.0\nCamera_width: 64\nCamera_height: 64\n")
    for bad in ("f_%s.pgm", "f_%d_%d.pgm", "f_%n.pgm", "f_%5000d.pgm", "f_%", "f_%x.pgm"):
        r = _cli("--vocabulary_file=v.txt", "--camera_settings=" + settings, "--in_video=" + os.path.join(str(tmp_path), bad),
                 "--out_dir=" + str(tmp_path), "--novisualize")
        assert r.returncode != 0 and "input video opens" in r.stderr, (bad, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,rot,vf,hf,camera_rgb", [("ppm", 90, False, True, 1), ("rgb", 180, True, False, 0), ("rgb", 0, False, False, 1)])
def test_cli_reads_rgb_frames_and_ingests_on_the_device(tmp_path, oracle, kind, rot, vf, hf, camera_rgb):
    """RGB24 frames as the reference's reader decodes them (image_sequence_reader.cc:138-208): a PPM sequence or a raw
    .rgb file goes into the page-locked slots as read; rotation (:186-205), --vertical_flip / --horizontal_flip (:53-58)
    and cvtColor by Camera_RGB (Tracking.cc:247-260) run on the device.  Every frame's keypoints and descriptors
    against the oracle on the oracle-ingested grey frame."""
    from pilotguru_amd import vocab as V
    from pilotguru_amd.synth import synth_ride
    w, h, nfr, nf = 400, 300, 5, 600
    ride = synth_ride(14, w, h, nfr, dx=4, dy=2)
    rgb = np.ascontiguousarray(np.stack([ride, np.roll(ride, 5, axis=2), 255 - ride], axis=3))
    d = str(tmp_path)
    if kind == "ppm":
        for i in range(nfr):
            with open(os.path.join(d, "%04d.ppm" % i), "wb") as f:
                f.write(b"P6\n%d %d\n255\n" % (w, h) + rgb[i].tobytes())
        video = os.path.join(d, "%04d.ppm")
    else:
        video = os.path.join(d, "ride.rgb")
        rgb.tofile(video)
    with open(os.path.join(d, "cam.yml"), "w") as f:
        f.write("%%YAML:1.0\n---\nCamera_width: %d\nCamera_height: %d\nCamera_fps: 30.\nCamera_RGB: %d\nORBextractor_nFeatures: %d\n" % (w, h, camera_rgb, nf))
    desc, weight, parent = V.synth_vocabulary(4, 3, seed=3)
    V.write_vocabulary_text(os.path.join(d, "voc.txt"), 4, 3, desc, weight, parent)
    args = ["--vocabulary_file=" + os.path.join(d, "voc.txt"), "--camera_settings=" + os.path.join(d, "cam.yml"), "--in_video=" + video,
            "--out_dir=" + d, "--novisualize", "--batch=2", "--rotation=%d" % rot, "--dump_features=" + os.path.join(d, "feat.bin")]
    if vf: args.append("--vertical_flip")
    if hf: args.append("--horizontal_flip")
    r = _cli(*args)
    assert r.returncode == 0, r.stderr
    raw = open(os.path.join(d, "feat.bin"), "rb").read()
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    off = 0
    for i in range(nfr):
        up = oracle.ingest_geometry(rgb[i], rot, vf, hf)
        gray = oracle.rgb_to_gray(np.ascontiguousarray(up if camera_rgb else up[:, :, ::-1]))
        okp, odesc = ora.extract(gray)
        fid, n = np.frombuffer(raw, np.int32, 2, off); off += 8
        assert fid == i and n == len(okp) > 300
        assert raw[off:off + 28 * n] == okp.tobytes(); off += 28 * n
        assert raw[off:off + 32 * n] == odesc.tobytes(); off += 32 * n
    assert off == len(raw)
    assert _cli(*(args + ["--rotation=45"])).returncode != 0


@pytest.mark.gpu
def test_cli_front_end_run_matches_python_path(tmp_path, oracle):
    import pilotguru_amd as pg
    from pilotguru_amd import vocab as V
    from pilotguru_amd.synth import synth_ride
    w, h, nfr, nf = 480, 360, 5, 800
    ride = synth_ride(12, w, h, nfr, dx=4, dy=2)
    d = str(tmp_path)
    for i in range(nfr):
        with open(os.path.join(d, "%06d.pgm" % i), "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (w, h) + ride[i].tobytes())
    with open(os.path.join(d, "cam.yml"), "w") as f:                 # keys as written by calibrate.cc:504-544
        f.write("%YAML:1.0\n---\nCamera_fps: 25.\nORBextractor_nFeatures: 800\nORBextractor_scaleFactor: 1.2\n"
                "ORBextractor_nLevels: 8\nORBextractor_iniThFAST: 20\nORBextractor_minThFAST: 7\n")
    desc, weight, parent = V.synth_vocabulary(5, 4, seed=3)
    V.write_vocabulary_text(os.path.join(d, "voc.txt"), 5, 4, desc, weight, parent)
    r = _cli("--vocabulary_file=" + os.path.join(d, "voc.txt"), "--camera_settings=" + os.path.join(d, "cam.yml"),
             "--in_video=" + os.path.join(d, "%06d.pgm"), "--out_dir=" + d, "--novisualize", "--batch=2",
             "--dump_features=" + os.path.join(d, "feat.bin"))
    assert r.returncode == 0, r.stderr
    out = json.load(open(os.path.join(d, "frontend-0.json")))
    assert [fr["frame_id"] for fr in out["frames"]] == list(range(nfr))
    assert out["frames"][2]["time_usec"] == 80000                    # 2 / 25 fps
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    voc = V.ORBVocabulary(text_file=os.path.join(d, "voc.txt"))
    voc.upload(ext)
    raw = open(os.path.join(d, "feat.bin"), "rb").read()
    off, prevF = 0, None
    for i in range(nfr):
        fid, n = np.frombuffer(raw, np.int32, 2, off); off += 8
        F = pg.Frame(ext, ride[i])
        assert fid == i and n == F.N == out["frames"][i]["n_keypoints"]
        assert raw[off:off + 28 * n] == F.mvKeys.tobytes(); off += 28 * n
        assert raw[off:off + 32 * n] == F.mDescriptors.tobytes(); off += 32 * n
        (bid, bval), fv = voc.transform(F.mDescriptors, 4)
        assert out["frames"][i]["n_bow_words"] == len(bid) and out["frames"][i]["n_feature_nodes"] == len(fv[0])
        if prevF is not None:
            prev = np.stack([prevF.mvKeys["x"], prevF.mvKeys["y"]], 1).astype(np.float32)
            nm, _ = pg.ORBmatcher(0.9, True).SearchForInitialization(prevF, F, prev, 100)
            assert out["frames"][i]["n_matches_prev"] == nm and nm > 50
        else:
            assert out["frames"][i]["n_matches_prev"] == -1
        prevF = F


@pytest.mark.gpu
def test_cli_config1_ride_against_the_oracle(tmp_path, oracle):
    """BASELINE.json configs[0]'s shape through the real CLI: a 10 s 640x480 clip (300 frames at 30 fps), 1000
    features.  Every frame of --dump_features (cv::KeyPoint records, descriptors) and every per-frame count
    of frontend-0.json against the ORACLE (not the Python binding of the same library): extraction
    (ORBextractor.cc:1042-1104), Frame::ComputeBoW (Frame.cc:399-406), MonocularInitialization's
    SearchForInitialization against the previous frame (Tracking.cc:596-597, ORBmatcher.cc:407-522); then the
    tail of TrackImageSequence (src/slam/track_image_sequence.cc:63-109) on poses stamped with the run's own
    frame times, against the oracle's post-processing."""
    import sys
    sys.path.insert(0, HERE)
    from _oracle_pool import oracle_ride
    from pilotguru_amd import vocab as V
    from pilotguru_amd.synth import synth_ride
    from oracle import orb_oracle
    w, h, nfr, nf = 640, 480, 300, 1000
    ride = synth_ride(33, w, h, nfr, dx=1, dy=0)
    d = str(tmp_path)
    ride.tofile(os.path.join(d, "clip.gray"))
    with open(os.path.join(d, "cam.yml"), "w") as f:
        f.write("%YAML:1.0\n---\nCamera_width: 640\nCamera_height: 480\nCamera_fps: 30.\nORBextractor_nFeatures: 1000\n"
                "ORBextractor_scaleFactor: 1.2\nORBextractor_nLevels: 8\nORBextractor_iniThFAST: 20\nORBextractor_minThFAST: 7\n")
    desc, weight, parent = V.synth_vocabulary(6, 4, seed=5)
    V.write_vocabulary_text(os.path.join(d, "voc.txt"), 6, 4, desc, weight, parent)
    r = _cli("--vocabulary_file=" + os.path.join(d, "voc.txt"), "--camera_settings=" + os.path.join(d, "cam.yml"),
             "--in_video=" + os.path.join(d, "clip.gray"), "--out_dir=" + d, "--novisualize", "--batch=16",
             "--dump_features=" + os.path.join(d, "feat.bin"))
    assert r.returncode == 0, r.stderr
    out = json.load(open(os.path.join(d, "frontend-0.json")))
    assert len(out["frames"]) == nfr
    oext, _ = oracle_ride(list(ride), (nf, 1.2, 8, 20, 7), match=False)
    voc = orb_oracle.VocabOracle(os.path.join(d, "voc.txt"))
    raw = open(os.path.join(d, "feat.bin"), "rb").read()
    off, prev = 0, None
    bounds = (0.0, float(w), 0.0, float(h))
    for i in range(nfr):
        fid, n = np.frombuffer(raw, np.int32, 2, off); off += 8
        okp, odesc = oext[i]
        fr = out["frames"][i]
        assert fid == i == fr["frame_id"] and n * 28 == len(okp) and fr["n_keypoints"] == n
        assert fr["time_usec"] == int(round(i * 1e6 / 30.0))
        assert raw[off:off + 28 * n] == okp, "keypoints of frame %d" % i
        off += 28 * n
        assert raw[off:off + 32 * n] == odesc, "descriptors of frame %d" % i
        off += 32 * n
        kp = np.frombuffer(okp, orb_oracle.KEYPOINT_DTYPE)
        de = np.frombuffer(odesc, np.uint8).reshape(-1, 32)
        if i % 10 == 0:                                               # BoW of every 10th frame (the oracle's transform is slow)
            (bid, _), fv = voc.transform(de, 4)
            assert fr["n_bow_words"] == len(bid) and fr["n_feature_nodes"] == len(fv[0])
        if prev is not None:
            pk, pd = prev
            pm = np.stack([pk["x"], pk["y"]], 1).astype(np.float32)
            nm, _, _ = orb_oracle.search_for_initialization(pk, pd, kp, de, bounds, pm, 100, 0.9, True)
            assert fr["n_matches_prev"] == nm, "SearchForInitialization of frame %d" % i
        else:
            assert fr["n_matches_prev"] == -1
        prev = (kp, de)
    assert off == len(raw)
    # the tail of TrackImageSequence on poses carrying this run's frame ids and time stamps
    sys.path.insert(0, HERE)
    from test_trajectory_post import _ride
    t, q = _ride(4, nfr)
    poses = os.path.join(d, "poses.txt")
    with open(poses, "w") as f:
        for i, fr in enumerate(out["frames"]):
            f.write("%d 0 %d %s\n" % (fr["time_usec"], fr["frame_id"], " ".join(repr(float(v)) for v in list(t[i]) + list(q[i]))))
    r = _cli("--poses_in=" + poses, "--out_dir=" + d, "--rotation_smooth_sigma=3")
    assert r.returncode == 0, r.stderr
    tj = json.load(open(os.path.join(d, "trajectory-0.json")))
    qs = orb_oracle.smooth_heading_directions(q, 3)
    vec, val, _ = orb_oracle.trajectory_pca(t)
    dirs = orb_oracle.project_directions(qs, vec[:2])
    turn = orb_oracle.turn_angles(dirs)
    r15 = lambda x: float("%.15g" % x)
    assert len(tj["trajectory"]) == nfr and [[r15(v) for v in row] for row in vec[:2]] == tj["plane"]
    for i, p in enumerate(tj["trajectory"]):
        assert p["frame_id"] == i and p["time_usec"] == out["frames"][i]["time_usec"]
        assert p["planar_direction"] == [r15(dirs[i, 0]), r15(dirs[i, 1])]
        if i:
            dt = (out["frames"][i]["time_usec"] - out["frames"][i - 1]["time_usec"]) * 1e-6
            assert p["angular_velocity"] == r15(turn[i] / (dt + 1e-10))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gray", "pgm", "y4m"])
def test_cli_shard_reproduces_the_unsharded_ride(tmp_path, kind):
    """--shard=rank/world (single ride over several GPUs from C++, one process per GPU): contiguous
    chunks with a one-frame overlap, nothing exchanged.  The shards' frontend-<rank>.json entries and --dump_features
    records, concatenated in rank order, equal the unsharded run -- including n_matches_prev of the first frame of every
    chunk (its predecessor lives in the previous chunk and is extracted twice) and the whole-ride frame ids / time stamps
    (src/slam/track_image_sequence.cc:43-52)."""
    from pilotguru_amd import vocab as V
    from pilotguru_amd.synth import synth_ride
    w, h, nfr, world = 320, 240, 11, 3
    ride = synth_ride(17, w, h, nfr, dx=3, dy=1)
    d = str(tmp_path)
    if kind == "gray":
        video = os.path.join(d, "clip.gray")
        ride.tofile(video)
    elif kind == "pgm":
        video = os.path.join(d, "f%04d.pgm")
        for i in range(nfr):
            with open(video % i, "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (w, h) + ride[i].tobytes())
    else:
        video = os.path.join(d, "clip.y4m")
        with open(video, "wb") as f:
            f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 Cmono\n" % (w, h))
            for i in range(nfr):
                f.write(b"FRAME\n" + ride[i].tobytes())
    with open(os.path.join(d, "cam.yml"), "w") as f:
        f.write("%%YAML:1.0\n---\nCamera_width: %d\nCamera_height: %d\nCamera_fps: 30.\nORBextractor_nFeatures: 500\n" % (w, h))
    desc, weight, parent = V.synth_vocabulary(5, 3, seed=9)
    V.write_vocabulary_text(os.path.join(d, "voc.txt"), 5, 3, desc, weight, parent)
    common = ["--vocabulary_file=" + os.path.join(d, "voc.txt"), "--camera_settings=" + os.path.join(d, "cam.yml"),
              "--in_video=" + video, "--novisualize", "--batch=4"]
    whole = os.path.join(d, "whole"); os.mkdir(whole)
    r = _cli(*common, "--out_dir=" + whole, "--dump_features=" + os.path.join(whole, "feat.bin"))
    assert r.returncode == 0, r.stderr
    want = json.load(open(os.path.join(whole, "frontend-0.json")))["frames"]
    assert [fr["frame_id"] for fr in want] == list(range(nfr))
    got, dump = [], b""
    for rank in range(world):
        od = os.path.join(d, "s%d" % rank); os.mkdir(od)
        r = _cli(*common, "--out_dir=" + od, "--shard=%d/%d" % (rank, world), "--dump_features=" + os.path.join(od, "feat.bin"))
        assert r.returncode == 0, r.stderr
        got += json.load(open(os.path.join(od, "frontend-%d.json" % rank)))["frames"]
        dump += open(os.path.join(od, "feat.bin"), "rb").read()
    assert got == want
    assert dump == open(os.path.join(whole, "feat.bin"), "rb").read()
    # --devices: the same shards as host threads of ONE process, the vocabulary parsed once and broadcast over RCCL behind
    # the C ABI (one rank per distinct device; contexts sharing a device copy from the rank's buffer) -- the merged report
    # and dump ARE the unsharded run's.  "0" = a one-device RCCL group; "0,0,0" = three contexts on the one GPU a test box has
    for devs, nctx in (("0", 1), ("0,0,0", 3), ("0,0,0,0,0", 5)):
        od = os.path.join(d, "dev" + str(nctx)); os.mkdir(od)
        r = _cli(*common, "--out_dir=" + od, "--devices=" + devs, "--dump_features=" + os.path.join(od, "feat.bin"))
        assert r.returncode == 0, r.stderr
        assert "parsed once, broadcast to %d context(s) on 1 device(s) over RCCL" % nctx in r.stderr, r.stderr
        assert open(os.path.join(od, "frontend-0.json")).read() == open(os.path.join(whole, "frontend-0.json")).read()
        assert open(os.path.join(od, "feat.bin"), "rb").read() == open(os.path.join(whole, "feat.bin"), "rb").read()
    assert _cli(*common, "--out_dir=" + od, "--devices=0,0", "--shard=0/2").returncode != 0      # one or the other
    assert _cli(*common, "--out_dir=" + od, "--devices=").returncode != 0
    # more shards than frames: the surplus ranks own nothing and say so
    od = os.path.join(d, "empty"); os.mkdir(od)
    r = _cli(*common, "--out_dir=" + od, "--shard=12/13")
    assert r.returncode == 0 and json.load(open(os.path.join(od, "frontend-12.json")))["frames"] == []
    assert _cli(*common, "--out_dir=" + od, "--shard=3/3").returncode != 0

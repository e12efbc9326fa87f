"""Trajectory post-processing: pilotguru_amd/csrc/post.cc through the C ABI against
the C oracle (oracle/post_oracle.c), bit for bit, plus order-independent cross-checks against
scipy / numpy (the OpenCV summation orders both sides restate are unpinned -- see the oracle's
header -- so the cross-checks bound how far a wrong order could move a result: ~1e-15 relative).

Host arithmetic only: runs without a GPU."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from pilotguru_amd import trajectory as T

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(ROOT, "pilotguru_amd", "host", "optical_trajectories")


def _ride(seed, n, vertical=1e-3):
    """A car-like trajectory: mostly planar translations, rotations mostly about one axis."""
    r = np.random.default_rng(seed)
    heading = np.cumsum(r.normal(0, 0.02, n))
    step = 0.5 + 0.1 * r.random(n)
    xy = np.cumsum(np.stack([np.cos(heading) * step, np.sin(heading) * step], 1), 0)
    t = np.stack([xy[:, 0], vertical * r.normal(0, 1, n), xy[:, 1]], 1)
    # tilt the whole thing so the plane is not axis aligned
    a = 0.3
    R = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    t = t @ R.T
    q = np.stack([np.cos(heading / 2), 0.01 * r.normal(0, 1, n), np.sin(heading / 2), 0.01 * r.normal(0, 1, n)], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return t, q


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.mark.parametrize("n,sigma", [(1, 1), (2, 1), (3, 2), (7, 1), (8, 3), (50, 1), (50, 5), (333, 2), (1001, 10)])
def test_smooth_heading_directions_equals_oracle(oracle, n, sigma):
    _, q = _ride(n * 7 + sigma, n)
    got = T.SmoothHeadingDirections(q, sigma)
    want = oracle.smooth_heading_directions(q, sigma)
    assert np.array_equal(_bits(got), _bits(want))
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-15)


def test_smooth_heading_directions_against_scipy():
    from scipy.ndimage import correlate1d
    _, q = _ride(5, 400)
    sigma = 4
    x = np.arange(4 * sigma + 1) - 2 * sigma
    k = np.exp(-0.5 * x * x / (sigma * sigma)); k /= k.sum()
    s = correlate1d(q, k, axis=0, mode="nearest")
    s /= np.linalg.norm(s, axis=1, keepdims=True)
    assert np.allclose(T.SmoothHeadingDirections(q, sigma), s, rtol=0, atol=1e-14)


def test_smooth_heading_directions_rejects_bad_sigma():
    from pilotguru_amd._lib import PgorbError
    _, q = _ride(1, 10)
    for s in (0, -1):
        with pytest.raises(PgorbError):                     # CHECK_GT(sigma, 0), smoothing.cc:14
            T.SmoothHeadingDirections(q, s)
    assert T.SmoothHeadingDirections(np.zeros((0, 4)), 2).shape == (0, 4)


@pytest.mark.parametrize("n,m,sigma", [(1, 4, 0.5), (2, 3, 0.003), (100, 100, 0.003), (500, 77, 0.05), (500, 500, 1.5)])
def test_smooth_time_series_equals_oracle(oracle, n, m, sigma):
    r = np.random.default_rng(n + m)
    times = np.cumsum(0.002 + 0.004 * r.random(n))
    values = r.normal(10, 3, n)
    targets = times if m == n else np.sort(r.uniform(times[0] - 0.1, times[-1] + 0.1, m))
    got = T.SmoothTimeSeries(values, times, targets, sigma)
    want = oracle.smooth_time_series(values, times, targets, sigma)
    assert np.array_equal(_bits(got), _bits(want))


def test_smooth_time_series_properties():
    # a constant series stays constant (the weights are a partition of [0, 1])
    t = np.linspace(0, 1, 200)
    assert np.allclose(T.SmoothTimeSeries(np.full(200, 3.25), t, t, 0.01), 3.25, rtol=0, atol=1e-14)
    # pure-Python restatement of smoothing.cc:57-97
    r = np.random.default_rng(3)
    t = np.cumsum(r.random(60) * 0.01); v = r.normal(0, 1, 60); sigma = 0.02
    cdf = lambda x, mu: 0.5 * (1.0 + math.erf((x - mu) / (math.sqrt(2.0) * sigma)))
    out, lo, hi = [], 0, 0
    for tt in t:
        while lo + 1 < 60 and (tt - t[lo + 1]) > 3 * sigma: lo += 1
        while hi + 1 < 60 and (t[hi] - tt) < 3 * sigma: hi += 1
        prev, acc = 0.0, 0.0
        for i in range(lo, hi):
            c = cdf((t[i] + t[i + 1]) / 2.0, tt)
            acc += v[i] * (c - prev); prev = c
        out.append(acc + v[hi] * (1.0 - prev))
    assert np.array_equal(_bits(T.SmoothTimeSeries(v, t, t, sigma)), _bits(np.array(out)))
    from pilotguru_amd._lib import PgorbError
    with pytest.raises(PgorbError):
        T.SmoothTimeSeries(v, t, t, 0.0)
    with pytest.raises(ValueError):
        T.SmoothTimeSeries(v[:-1], t, t, 1.0)


@pytest.mark.parametrize("n", [3, 4, 5, 6, 7, 8, 9, 100, 1234])
def test_trajectory_pca_equals_oracle(oracle, n):
    t, _ = _ride(n, n)
    vec, val, mean = T.TrajectoryToPCA(t)
    ovec, oval, omean = oracle.trajectory_pca(t)
    assert np.array_equal(_bits(vec), _bits(ovec)) and np.array_equal(_bits(val), _bits(oval))
    assert np.array_equal(_bits(mean), _bits(omean))


def test_trajectory_pca_against_numpy():
    t, _ = _ride(11, 800)
    vec, val, mean = T.TrajectoryToPCA(t)
    assert np.allclose(mean, t.mean(0), rtol=1e-13)
    c = np.cov(t.T, bias=True)                                   # CV_COVAR_SCALE: 1/n
    w, v = np.linalg.eigh(c)
    assert np.allclose(val, w[::-1], rtol=1e-9, atol=1e-12 * w[-1])
    for k in range(3):                                           # eigenvectors as rows, sign free
        assert min(np.abs(vec[k] - v[:, 2 - k]).max(), np.abs(vec[k] + v[:, 2 - k]).max()) < 1e-7
    assert np.allclose(vec @ vec.T, np.eye(3), atol=1e-14)
    assert val[0] >= val[1] >= val[2] >= 0
    from pilotguru_amd._lib import PgorbError
    with pytest.raises(PgorbError):
        T.TrajectoryToPCA(t[:2])


@pytest.mark.parametrize("n", [1, 2, 17, 500])
def test_projection_and_turn_angles_equal_oracle(oracle, n):
    t, q = _ride(n + 40, max(n, 3))
    vec, _, _ = T.TrajectoryToPCA(t)
    t, q = t[:n], q[:n]
    plane = vec[:2]
    d = T.ProjectDirections(q, plane)
    assert np.array_equal(_bits(d), _bits(oracle.project_directions(q, plane)))
    assert np.array_equal(_bits(T.ProjectTranslations(t, plane)), _bits(oracle.project_translations(t, plane)))
    a = T.Projected2DDirectionsToTurnAngles(d)
    assert np.array_equal(_bits(a), _bits(oracle.turn_angles(d)))
    assert a[0] == 0


def test_projection_geometry():
    # rotation about y by angle h maps the optical axis (0,0,1) to (sin h, 0, cos h)
    h = np.array([0.0, 0.1, 0.25, 0.2, -0.4])
    q = np.stack([np.cos(h / 2), np.zeros(5), np.sin(h / 2), np.zeros(5)], 1)
    plane = np.array([[1.0, 0, 0], [0, 0, 1.0]])
    d = T.ProjectDirections(q, plane)
    assert np.allclose(d, np.stack([np.sin(h), np.cos(h)], 1), atol=1e-15)
    a = T.Projected2DDirectionsToTurnAngles(d)
    # cross_z = prev.x*curr.y - prev.y*curr.x: positive when the direction turns from x towards y
    assert np.allclose(np.abs(a[1:]), np.abs(np.diff(h)), atol=1e-7)
    assert list(np.sign(a[1:])) == list(-np.sign(np.diff(h)))
    # translations land in the plane: projecting twice changes nothing measurable
    t = np.random.default_rng(0).normal(0, 1, (20, 3))
    p1 = T.ProjectTranslations(t, plane)
    assert np.allclose(p1[:, 1], 0) and np.allclose(T.ProjectTranslations(p1, plane), p1, atol=1e-15)


def test_flatten_trajectory_gate():
    t, q = _ride(2, 300)
    out = T.FlattenTrajectory(t, q, rotation_smooth_sigma=3)
    assert out is not None and out["plane"].shape == (2, 3) and len(out["turn_angles"]) == 300
    t2, q2 = _ride(2, 300, vertical=5.0)                          # vertical motion not negligible: dropped (:83-90)
    assert T.FlattenTrajectory(t2, q2) is None


def test_cli_poses_in_writes_the_reference_json(tmp_path, oracle):
    """--poses_in runs track_image_sequence.cc:63-109; the file must hold exactly what the oracle's
    functions + the JSON schema give."""
    n = 40
    t, q = _ride(9, n)
    poses = os.path.join(str(tmp_path), "poses.txt")
    with open(poses, "w") as f:
        for i in range(n):
            f.write("%d %d %d %s\n" % (1000000 + 33333 * i, 0, i, " ".join(repr(float(v)) for v in list(t[i]) + list(q[i]))))
    r = subprocess.run([CLI, "--poses_in=" + poses, "--out_dir=" + str(tmp_path), "--rotation_smooth_sigma=2", "--segment_id=3"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(os.path.join(str(tmp_path), "trajectory-3.json")))
    qs = oracle.smooth_heading_directions(q, 2)
    vec, val, _ = oracle.trajectory_pca(t)
    dirs = oracle.project_directions(qs, vec[:2])
    turn = oracle.turn_angles(dirs)
    r15 = lambda x: float("%.15g" % x)                           # the writer prints 15 significant digits
    assert [[r15(v) for v in row] for row in vec[:2]] == d["plane"]
    for i, p in enumerate(d["trajectory"]):
        assert p["frame_id"] == i and p["time_usec"] == 1000000 + 33333 * i and p["is_lost"] is False
        assert p["planar_direction"] == [r15(dirs[i, 0]), r15(dirs[i, 1])]
        assert [p["pose"]["rotation"][k] for k in "wxyz"] == [r15(v) for v in qs[i]]
        assert p["pose"]["translation"] == [r15(v) for v in t[i]]
        if i:
            assert p["angular_velocity"] == r15(turn[i] / (33333 * 1e-6 + 1e-10))
    # dropped trajectory: no file, exit 0, the reference's warning
    t2, q2 = _ride(2, 300, vertical=5.0)
    with open(poses, "w") as f:
        for i in range(300):
            f.write("%d 0 %d %s\n" % (i, i, " ".join(repr(float(v)) for v in list(t2[i]) + list(q2[i]))))
    r = subprocess.run([CLI, "--poses_in=" + poses, "--out_dir=" + str(tmp_path), "--segment_id=4"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 0 and "3rd eigenvalue was too large" in r.stderr
    assert not os.path.exists(os.path.join(str(tmp_path), "trajectory-4.json"))

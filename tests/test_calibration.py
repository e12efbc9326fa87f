"""fit_motion's velocity calibration (BASELINE configs[4]).

CPU part: the oracle (oracle/calib_oracle.c) against what can be known without the reference binary --
the merge example of align_time_series.hpp, finite-difference gradients, L-BFGS actually descending,
a recoverable synthetic calibration.  GPU part: pilotguru_amd/csrc/calib.hip (one lane per window)
against the oracle, bit for bit."""
import math

import numpy as np
import pytest


def imu_ride(seed, n_gps=30, imu_hz=50.0, noise=0.02):
    """A drive seen by a phone: GPS speed at ~1 Hz, gyroscope and accelerometer at ~imu_hz with their own
    clocks.  Device frame = world frame rotated by the integrated yaw; the accelerometer reads the
    world acceleration in the device frame minus gravity plus a bias."""
    r = np.random.default_rng(seed)
    T = float(n_gps + 1)
    t_rot = np.cumsum(r.uniform(0.8, 1.2, int(T * imu_hz)) / imu_hz) + 0.013
    t_acc = np.cumsum(r.uniform(0.8, 1.2, int(T * imu_hz)) / imu_hz) + 0.007
    t_rot, t_acc = t_rot[t_rot < T], t_acc[t_acc < T]
    yaw_rate = lambda t: 0.25 * np.sin(0.35 * t + seed)
    yaw = lambda t: -0.25 / 0.35 * (np.cos(0.35 * t + seed) - math.cos(seed))
    speed = lambda t: 12.0 + 4.0 * np.sin(0.2 * t + 0.5 * seed)
    dspeed = lambda t: 0.8 * np.cos(0.2 * t + 0.5 * seed)
    rot = np.stack([0.01 * r.normal(0, 1, len(t_rot)), 0.01 * r.normal(0, 1, len(t_rot)), yaw_rate(t_rot)], 1)
    rot += noise * 0.1 * r.normal(0, 1, rot.shape)
    # world acceleration of a point moving with speed(t) along heading yaw(t)
    h = yaw(t_acc)
    ax = dspeed(t_acc) * np.cos(h) - speed(t_acc) * yaw_rate(t_acc) * np.sin(h)
    ay = dspeed(t_acc) * np.sin(h) + speed(t_acc) * yaw_rate(t_acc) * np.cos(h)
    # into the device frame (rotation by -yaw about z), gravity-like global bias and a device bias on top
    acc = np.stack([ax * np.cos(h) + ay * np.sin(h), -ax * np.sin(h) + ay * np.cos(h), np.full(len(h), 9.81)], 1)
    acc += np.array([0.05, -0.03, 0.02]) + noise * r.normal(0, 1, acc.shape)
    t_gps = np.arange(1, n_gps + 1) * 1.0 + r.uniform(-0.05, 0.05, n_gps)
    gps_v = speed(t_gps) + noise * r.normal(0, 1, n_gps)
    us = lambda t: np.round(t * 1e6).astype(np.int64) + 1_500_000_000_000_000
    return (gps_v, us(t_gps)), (rot, us(t_rot)), (acc, us(t_acc))


def irregular_series(r):
    """Recorder series with everything irregular: 2-60 fixes at any rate, IMU clocks from 3 to 400 Hz with jitter,
    offsets against the GPS clock, gaps in the recording."""
    n_gps = int(r.integers(2, 60))
    gps_dt = r.uniform(0.2, 2.5)
    t_gps = np.cumsum(r.uniform(0.5, 1.5, n_gps) * gps_dt) + r.uniform(0, 3)
    T = t_gps[-1] + r.uniform(-1.0, 2.0)

    def imu(hz):
        n = max(3, int(T * hz))
        t = np.cumsum(r.uniform(0.3, 1.7, n) / hz) + r.uniform(-1.0, 1.5)
        if r.random() < 0.3:
            k = int(r.integers(1, n)); t[k:] += r.uniform(0.5, 3.0)
        return t
    t_rot, t_acc = imu(r.uniform(3, 400)), imu(r.uniform(3, 400))
    rot = r.normal(0, 0.3, (len(t_rot), 3)); acc = r.normal(0, 2.0, (len(t_acc), 3)) + [0, 0, 9.8]
    us = lambda t: np.unique(np.round(t * 1e6).astype(np.int64) + 10**15)
    tg, tr, ta = us(t_gps), us(t_rot), us(t_acc)
    return (np.abs(r.normal(10, 5, len(tg))), tg), (rot[:len(tr)], tr), (acc[:len(ta)], ta)


def _bits(a):
    """Bit patterns, all NaNs made one (a window without IMU samples is 0/0 in the reference; which NaN
    comes out depends on the instruction set, and the JSON prints none of them)."""
    a = np.ascontiguousarray(a, np.float64).copy()
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


# ---------------------------------------------------------------- CPU: the oracle

def test_oracle_gradient_matches_finite_differences(oracle):
    gps, rot, acc = imu_ride(1, n_gps=12)
    r = np.random.default_rng(0)
    for _ in range(3):
        x = r.normal(0, 0.5, 9); x[2] -= 9.8
        f0, g = oracle.calibrator_eval(*gps, *rot, *acc, x)
        assert np.isfinite(f0) and np.all(np.isfinite(g))
        for k in range(9):
            e = np.zeros(9); e[k] = 1e-6
            fp, _ = oracle.calibrator_eval(*gps, *rot, *acc, x + e)
            fm, _ = oracle.calibrator_eval(*gps, *rot, *acc, x - e)
            num = (fp - fm) / 2e-6
            # the reference's analytic gradient is approximate by design (the 1e-5 in the norm, the
            # "TODO: not quite right" travel rule): same sign and size, not the same digits
            assert abs(num - g[k]) <= 0.05 * max(abs(num), abs(g[k])) + 1e-4, (k, num, g[k])


def test_oracle_lbfgs_descends_and_is_deterministic(oracle):
    gps, rot, acc = imu_ride(2, n_gps=14)
    f0, _ = oracle.calibrator_eval(*gps, *rot, *acc, np.zeros(9))
    x, res, it = oracle.fit_windows(*gps, *rot, *acc, batch_size=8, shift_step=4, max_iters=60)
    assert len(x) == 4 and np.all(it >= 1) and np.all(it <= 60)
    assert np.all(res < f0) and np.all(np.isfinite(x))
    # the first window's fit evaluated again gives the residual the solver reported
    f, _ = oracle.calibrator_eval(gps[0][:8], gps[1][:8], *rot, *acc, x[0])
    assert f == res[0]
    x2, res2, it2 = oracle.fit_windows(*gps, *rot, *acc, batch_size=8, shift_step=4, max_iters=60)
    assert np.array_equal(_bits(x), _bits(x2)) and np.array_equal(it, it2)


def test_oracle_rejects_what_the_reference_checks(oracle):
    gps, rot, acc = imu_ride(3, n_gps=6)
    with pytest.raises(ValueError):                                   # CheckTimestampsIncreasing
        oracle.calibrator_eval(gps[0], gps[1][::-1].copy(), *rot, *acc, np.zeros(9))
    with pytest.raises(ValueError):                                   # disjoint series cannot be merged
        oracle.calibrator_eval(*gps, rot[0], rot[1] + 10**9, *acc, np.zeros(9))


def test_oracle_velocity_pipeline_recovers_the_speed(oracle):
    gps, rot, acc = imu_ride(4, n_gps=24, noise=0.0)
    t, v, fwd = oracle.fit_motion_velocities(*gps, *rot, *acc, [0.0, 0.0, 1.0], batch_size=10, shift_step=5, max_iters=200)
    assert len(t) > 500 and np.all(np.diff(t) > 0)
    ts = (t - 1_500_000_000_000_000) * 1e-6
    truth = 12.0 + 4.0 * np.sin(0.2 * ts + 0.5 * 4)
    assert np.median(np.abs(v - truth)) < 0.5                        # IMU-rate speed follows the GPS-rate truth
    assert abs(np.linalg.norm(fwd) - 1.0) < 1e-3 and abs(fwd[2]) < 1e-9 and fwd[0] > 0.9    # the car drives along device +x


# ---------------------------------------------------------------- GPU: calib.hip == oracle

@pytest.fixture(scope="module")
def ctx():
    import pilotguru_amd as pg
    return pg.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=1)


@pytest.mark.gpu
def test_gpu_calibrator_eval_equals_oracle(ctx, oracle):
    from pilotguru_amd.calibration import AccelerometerCalibrator
    gps, rot, acc = imu_ride(5, n_gps=9)
    cal = AccelerometerCalibrator(ctx, gps, rot, acc)
    r = np.random.default_rng(1)
    xs = np.concatenate([np.zeros((1, 9)), r.normal(0, 1, (70, 9))])           # more points than one wave
    fx, g = cal(xs)
    for i in range(len(xs)):
        f0, g0 = oracle.calibrator_eval(*gps, *rot, *acc, xs[i])
        assert _bits(fx[i]) == _bits(f0) and np.array_equal(_bits(g[i]), _bits(g0)), i


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_gps,batch,shift,iters", [(6, 20, 8, 4, 40), (7, 33, 10, 5, 500), (8, 7, 40, 5, 25), (9, 400, 12, 3, 30)])
def test_gpu_window_fits_equal_oracle(ctx, oracle, seed, n_gps, batch, shift, iters):
    from pilotguru_amd.calibration import FitVelocityWindows
    gps, rot, acc = imu_ride(seed, n_gps=n_gps, imu_hz=20.0 if n_gps > 100 else 50.0)
    x, res, it = FitVelocityWindows(ctx, gps, rot, acc, batch, shift, iters)
    ox, ores, oit = oracle.fit_windows(*gps, *rot, *acc, batch_size=batch, shift_step=shift, max_iters=iters)
    assert len(x) == math.ceil(n_gps / shift)
    assert np.array_equal(it, oit)
    assert np.array_equal(_bits(x), _bits(ox)) and np.array_equal(_bits(res), _bits(ores))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_irregular_series_equal_oracle(ctx, oracle, seed):
    """The committed slice of tools/experiments/fuzz_calib.py (750 cases there): short and long windows, single-fix
    windows, intervals without IMU samples, partial chunks of every length."""
    from pilotguru_amd.calibration import ComputeForwardVelocitiesFromImu, FitVelocityWindows
    r = np.random.default_rng(100 + seed)
    gps, rot, acc = irregular_series(r)
    batch = int(r.integers(1, 45)); shift = int(r.integers(1, batch + 1)); iters = int(r.integers(1, 60))
    ox, ores, oit = oracle.fit_windows(*gps, *rot, *acc, batch, shift, iters)
    x, res, it = FitVelocityWindows(ctx, gps, rot, acc, batch, shift, iters)
    assert np.array_equal(it, oit) and np.array_equal(_bits(x), _bits(ox)) and np.array_equal(_bits(res), _bits(ores))
    if np.all(oit >= 0):
        axis = np.array([0.1, -0.2, 1.0]); axis /= np.linalg.norm(axis)
        t, v, f = ComputeForwardVelocitiesFromImu(ctx, gps, rot, acc, axis, batch, shift, iters, 0.01, 3.0, 0.1)
        ot, ov, of = oracle.fit_motion_velocities(*gps, *rot, *acc, axis, batch, shift, iters, 0.01, 3.0, 0.1)
        assert np.array_equal(t, ot) and np.array_equal(_bits(v), _bits(ov)) and np.array_equal(_bits(f), _bits(of))


@pytest.mark.gpu
def test_gpu_fit_motion_velocities_equal_oracle(ctx, oracle):
    from pilotguru_amd.calibration import ComputeForwardVelocitiesFromImu
    gps, rot, acc = imu_ride(10, n_gps=30)
    axis = np.array([0.02, -0.01, 1.0]); axis /= np.linalg.norm(axis)
    t, v, fwd = ComputeForwardVelocitiesFromImu(ctx, gps, rot, acc, axis, 10, 5, 120, 0.01, 5.0, 0.2)
    ot, ov, ofwd = oracle.fit_motion_velocities(*gps, *rot, *acc, axis, 10, 5, 120, 0.01, 5.0, 0.2)
    assert np.array_equal(t, ot) and np.array_equal(_bits(v), _bits(ov)) and np.array_equal(_bits(fwd), _bits(ofwd))


@pytest.mark.gpu
def test_gpu_calibration_argument_errors(ctx):
    from pilotguru_amd._lib import PgorbError
    from pilotguru_amd.calibration import FitVelocityWindows
    gps, rot, acc = imu_ride(11, n_gps=8)
    with pytest.raises(PgorbError):                                   # CHECK_GE(batch, shift), fit_motion.cc:304
        FitVelocityWindows(ctx, gps, rot, acc, 4, 5, 10)
    with pytest.raises(PgorbError):
        FitVelocityWindows(ctx, (gps[0], gps[1][::-1].copy()), rot, acc, 8, 4, 10)
    with pytest.raises(PgorbError):
        FitVelocityWindows(ctx, gps, (rot[0], rot[1] + 10**9), acc, 8, 4, 10)


# ---------------------------------------------------------------- rotation.cc (host) and the fit_motion CLI

@pytest.mark.parametrize("seed,n_gps,interval", [(12, 12, 500000), (13, 40, 250000), (14, 6, 1000000)])
def test_principal_rotation_axes_and_steering_equal_oracle(oracle, seed, n_gps, interval):
    from pilotguru_amd.calibration import GetAngularVelocitiesAroundAxisDirect, GetPrincipalRotationAxes
    _, rot, _ = imu_ride(seed, n_gps=n_gps)
    vec = GetPrincipalRotationAxes(rot, interval)
    assert np.array_equal(_bits(vec), _bits(oracle.principal_rotation_axes(*rot, interval)))
    assert np.allclose(vec @ vec.T, np.eye(3), atol=1e-13)
    assert abs(abs(vec[0, 2]) - 1.0) < 0.05                          # the synthetic car only yaws: vertical = device z
    st = GetAngularVelocitiesAroundAxisDirect(rot, vec[0])
    assert np.array_equal(_bits(st), _bits(oracle.angular_velocities_around_axis(rot[0], vec[0])))
    assert np.allclose(st, rot[0] @ vec[0], atol=1e-14)


def test_principal_rotation_axes_against_numpy_and_errors():
    from pilotguru_amd._lib import PgorbError
    from pilotguru_amd.calibration import GetAngularVelocitiesAroundAxisDirect, GetPrincipalRotationAxes
    _, rot, _ = imu_ride(15, n_gps=30)
    r, t = rot
    # the integrated interval rotations, straight from the definition
    rows, q, acc_us = [], np.array([1.0, 0, 0, 0]), 0
    for i in range(1, len(r)):
        dt = (t[i] - t[i - 1]) * 1e-6
        rate = np.linalg.norm(r[i]); h = rate * dt * 0.5
        b = np.concatenate([[math.cos(h)], r[i] * (math.sin(h) / (rate + 1e-30))])
        q = np.array([q[0] * b[0] - q[1:] @ b[1:], *(q[0] * b[1:] + b[0] * q[1:] + np.cross(q[1:], b[1:]))])
        acc_us += t[i] - t[i - 1]
        if acc_us >= 500000:
            rows.append(q[1:].copy()); q = np.array([1.0, 0, 0, 0]); acc_us = 0
    w, v = np.linalg.eigh(np.cov(np.array(rows).T, bias=True))
    vec = GetPrincipalRotationAxes(rot)
    for k in range(3):
        assert min(np.abs(vec[k] - v[:, 2 - k]).max(), np.abs(vec[k] + v[:, 2 - k]).max()) < 1e-6
    with pytest.raises(PgorbError):                                   # CHECK_GT(integration_interval_usec, 0)
        GetPrincipalRotationAxes(rot, 0)
    with pytest.raises(PgorbError):                                   # CHECK_GE(interval_rotations.size(), 3)
        GetPrincipalRotationAxes((r[:20], t[:20]), 10**9)
    with pytest.raises(PgorbError):                                   # axis not normalised
        GetAngularVelocitiesAroundAxisDirect(rot, [0, 0, 2.0])


def _write_recorder_json(d, gps, rot, acc):
    import json
    import os
    def dump(name, root, rows):
        with open(os.path.join(d, name), "w") as f:
            json.dump({root: rows}, f)
    dump("locations.json", "locations", [{"speed_m_s": float(v), "time_usec": int(t), "lat": 1.0, "lon": 2.0} for v, t in zip(*gps)])
    dump("rotations.json", "rotations", [{"x": float(a[0]), "y": float(a[1]), "z": float(a[2]), "time_usec": int(t)} for a, t in zip(*rot)])
    dump("accelerations.json", "accelerations", [{"x": float(a[0]), "y": float(a[1]), "z": float(a[2]), "time_usec": int(t)} for a, t in zip(*acc)])


def test_fit_motion_cli_checks_flags(tmp_path):
    import os
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pilotguru_amd", "host", "fit_motion")
    assert os.path.exists(cli), "build it: make -C pilotguru_amd/csrc"
    run = lambda *a: subprocess.run([cli] + list(a), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    r = run()
    assert r.returncode != 0 and "!FLAGS_rotations_json.empty()" in r.stderr
    r = run("--rotations_json=a", "--accelerations_json=b", "--locations_json=c", "--locations_batch_size=3", "--locations_shift_step=5")
    assert r.returncode != 0 and "FLAGS_locations_batch_size >= FLAGS_locations_shift_step" in r.stderr
    r = run("--rotations_json=a", "--accelerations_json=b", "--locations_json=c", "--post_smoothing_sigma_sec=0")
    assert r.returncode != 0 and "FLAGS_post_smoothing_sigma_sec > 0" in r.stderr
    # the steering output needs no GPU: host arithmetic only
    gps, rot, acc = imu_ride(16, n_gps=10)
    _write_recorder_json(str(tmp_path), gps, rot, acc)
    d = str(tmp_path)
    r = run("--rotations_json=%s/rotations.json" % d, "--accelerations_json=%s/accelerations.json" % d, "--locations_json=%s/locations.json" % d,
            "--steering_out_json=%s/steering.json" % d)
    assert r.returncode == 0, r.stderr
    import json
    from pilotguru_amd.calibration import GetAngularVelocitiesAroundAxisDirect, GetPrincipalRotationAxes
    st = json.load(open(d + "/steering.json"))["steering"]
    want = GetAngularVelocitiesAroundAxisDirect(rot, GetPrincipalRotationAxes(rot)[0])
    assert [e["time_usec"] for e in st] == [int(t) for t in rot[1]]
    assert [e["angular_velocity"] for e in st] == [float("%.15g" % v) for v in want]
    txt = open(d + "/steering.json").read()
    assert txt.startswith('{\n  "steering": [\n    {\n      "angular_velocity": ') and txt.endswith("\n  ]\n}\n")


@pytest.mark.gpu
def test_gpu_fit_motion_cli_writes_the_oracles_numbers(tmp_path, oracle):
    import json
    import os
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pilotguru_amd", "host", "fit_motion")
    gps, rot, acc = imu_ride(17, n_gps=26)
    d = str(tmp_path)
    _write_recorder_json(d, gps, rot, acc)
    r = subprocess.run([cli, "--rotations_json=%s/rotations.json" % d, "--accelerations_json=%s/accelerations.json" % d,
                        "--locations_json=%s/locations.json" % d, "--velocities_out_json=%s/v.json" % d, "--forward_axis_out_json=%s/f.json" % d,
                        "--steering_out_json=%s/s.json" % d, "--locations_batch_size=10", "--optimization_iters=80",
                        "--post_smoothing_sigma_sec=0.01"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 0, r.stderr
    axis = oracle.principal_rotation_axes(*rot)[0]
    ot, ov, ofwd = oracle.fit_motion_velocities(*gps, *rot, *acc, axis, 10, 5, 80, 0.01, 5.0, 0.2)
    r15 = lambda x: float("%.15g" % x)
    v = json.load(open(d + "/v.json"))["velocities"]
    assert [e["time_usec"] for e in v] == [int(t) for t in ot]
    assert [e["speed_m_s"] for e in v] == [r15(x) for x in ov]
    f = json.load(open(d + "/f.json"))["forward_axis"]
    assert [f["x"], f["y"], f["z"]] == [r15(x) for x in ofwd]
    s = json.load(open(d + "/s.json"))["steering"]
    assert [e["angular_velocity"] for e in s] == [r15(x) for x in oracle.angular_velocities_around_axis(rot[0], axis)]


def test_kahan_sum_equals_the_reference_header(oracle):
    """include/math/math.hpp is the one file of the fit_motion path that builds with this image's toolchain:
    oracle/_ref/libmath_ref.so is the reference's own KahanSum; the oracle's and the product's must agree with it
    bit for bit (and differ from the naive sum on an ill-conditioned series, or the test tests nothing)."""
    from pilotguru_amd.calibration import KahanSum
    r = np.random.default_rng(6)
    v = r.normal(0, 1, (5000, 3)) * 10.0 ** r.integers(-8, 9, (5000, 1))
    ref = oracle.ref_kahan_sum(v)
    if ref is None:
        pytest.skip("oracle/_ref/libmath_ref.so not built and /root/reference absent")
    assert np.array_equal(_bits(oracle.kahan_sum(v)), _bits(ref))
    assert np.array_equal(_bits(KahanSum(v)), _bits(ref))
    naive = np.zeros(3)
    for row in v:
        naive = naive + row
    assert not np.array_equal(_bits(naive), _bits(ref))
    assert np.array_equal(_bits(KahanSum(np.zeros((0, 3)))), _bits(np.zeros(3)))

"""Latency of ONE AccelerometerCalibrator evaluation on the GPU (one wave) and the throughput of many.
PGORB_CALIB_TIMING=1 python tools/experiments/calib_eval_latency.py [imu_hz]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_calibration import imu_ride  # noqa: E402

import pilotguru_amd as pg  # noqa: E402
from pilotguru_amd.calibration import AccelerometerCalibrator  # noqa: E402
from oracle import orb_oracle as orc  # noqa: E402

hz = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
gps, rot, acc = imu_ride(31, n_gps=40, imu_hz=hz)
ctx = pg.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=1)
cal = AccelerometerCalibrator(ctx, gps, rot, acc)
r = np.random.default_rng(0)
for n in (1, 1, 256, 1024, 4096):
    xs = r.normal(0, 1, (n, 9))
    t = time.time(); cal(xs); dt = time.time() - t
    print("n_points %5d: %.3f ms wall" % (n, dt * 1e3), flush=True)
x = r.normal(0, 1, 9)
t = time.time()
for _ in range(20):
    orc.calibrator_eval(*gps, *rot, *acc, x)
print("CPU oracle: %.3f ms per evaluation (incl. building the calibrator)" % ((time.time() - t) / 20 * 1e3))

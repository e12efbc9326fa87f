"""The whole velocity pipeline of fit_motion (window fits on the GPU + integration, averaging, smoothing, forward axis on
the host) on a synthetic ride, timed, and -- with a second argument -- compared with the CPU oracle bit for bit.
usage: python tools/experiments/fit_motion_rate.py [n_gps] [check]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_calibration import imu_ride  # noqa: E402

import pilotguru_amd as pg  # noqa: E402
from pilotguru_amd.calibration import ComputeForwardVelocitiesFromImu, GetPrincipalRotationAxes  # noqa: E402

n_gps = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
gps, rot, acc = imu_ride(21, n_gps=n_gps, imu_hz=100.0)
ctx = pg.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=1)
axis = GetPrincipalRotationAxes(rot)[0]
ComputeForwardVelocitiesFromImu(ctx, (gps[0][:50], gps[1][:50]), rot, acc, axis, 40, 5, 5)      # warm-up
t = time.time()
tt, v, fwd = ComputeForwardVelocitiesFromImu(ctx, gps, rot, acc, axis)
t_gpu = time.time() - t
print("fit_motion velocities: %d GPS fixes, %d output samples, %.2f s end to end; forward axis %s" % (n_gps, len(tt), t_gpu, np.round(fwd, 4)))
if len(sys.argv) > 2:
    from oracle import orb_oracle as orc
    t = time.time()
    ot, ov, of = orc.fit_motion_velocities(*gps, *rot, *acc, axis)
    t_cpu = time.time() - t
    same = np.array_equal(tt, ot) and np.array_equal(v.view(np.uint64), ov.view(np.uint64)) and np.array_equal(fwd.view(np.uint64), of.view(np.uint64))
    print("CPU oracle: %.1f s -> %.0fx; outputs bit-equal: %s" % (t_cpu, t_cpu / t_gpu, same))

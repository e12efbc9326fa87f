"""fit_motion velocity calibration: GPU (all windows at once) vs the CPU oracle (window after window).
usage: python tools/experiments/calib_rate.py [n_gps] [imu_hz] [cpu_windows]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_calibration import imu_ride  # noqa: E402

import pilotguru_amd as pg  # noqa: E402
from pilotguru_amd.calibration import FitVelocityWindows  # noqa: E402
from oracle import orb_oracle as orc  # noqa: E402

n_gps = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
hz = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
cpu_w = int(sys.argv[3]) if len(sys.argv) > 3 else 8
gps, rot, acc = imu_ride(21, n_gps=n_gps, imu_hz=hz)
ctx = pg.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=1)
FitVelocityWindows(ctx, (gps[0][:50], gps[1][:50]), rot, acc, 40, 5, 5)       # warm-up (module load)
t = time.time()
x, res, it = FitVelocityWindows(ctx, gps, rot, acc, 40, 5, 500)
t_gpu = time.time() - t
print("GPU: %d windows, %d gyro + %d accel samples, %.2f s total (host prep + upload + solver), iterations min/median/max %d/%d/%d"
      % (len(x), len(rot[0]), len(acc[0]), t_gpu, it.min(), int(np.median(it)), it.max()))
# CPU oracle on the first cpu_w windows' worth of GPS fixes (windows [0, cpu_w) are unaffected by the cut only when complete)
m = len(gps[0]) if cpu_w <= 0 else 40 + 5 * (cpu_w - 1)
cpu_w = len(x) if cpu_w <= 0 else cpu_w
t = time.time()
ox, ores, oit = orc.fit_windows(gps[0][:m], gps[1][:m], *rot, *acc, 40, 5, 500)
t_cpu = time.time() - t
same = np.array_equal(ox[:cpu_w].view(np.uint64), x[:cpu_w].view(np.uint64)) and np.array_equal(oit[:cpu_w], it[:cpu_w])
evals_cpu = t_cpu / len(ox)
print("CPU oracle: %d windows in %.2f s = %.3f s/window -> %.1f s for all %d windows; GPU/CPU = %.1fx; first %d windows bit-equal: %s"
      % (len(ox), t_cpu, evals_cpu, evals_cpu * len(x), len(x), evals_cpu * len(x) / t_gpu, cpu_w, same))

"""Random irregular recorder series through K9 (pilotguru_amd/csrc/calib.hip) and the CPU oracle: parameters,
residuals, iteration counts and the velocity pipeline must agree bit for bit (NaNs as NaNs).
usage: python tools/experiments/fuzz_calib.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pilotguru_amd as pg  # noqa: E402
from pilotguru_amd.calibration import ComputeForwardVelocitiesFromImu, FitVelocityWindows  # noqa: E402
from oracle import orb_oracle as orc  # noqa: E402


def bits(a):
    a = np.ascontiguousarray(a, np.float64).copy()
    a[np.isnan(a)] = np.nan
    return a.view(np.uint64)


sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_calibration import irregular_series as series  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
r = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pg.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=1)
done = skipped = 0
for c in range(cases):
    gps, rot, acc = series(r)
    batch = int(r.integers(1, 45)); shift = int(r.integers(1, batch + 1)); iters = int(r.integers(1, 60))
    try:
        ox, ores, oit = orc.fit_windows(*gps, *rot, *acc, batch, shift, iters)
    except ValueError:
        skipped += 1                                # the reference CHECK-fails on this input (disjoint series, ...)
        try:
            FitVelocityWindows(ctx, gps, rot, acc, batch, shift, iters)
            raise SystemExit("case %d: the oracle rejects the input, the product does not" % c)
        except pg._lib.PgorbError:
            continue
    x, res, it = FitVelocityWindows(ctx, gps, rot, acc, batch, shift, iters)
    ok = np.array_equal(it, oit) and np.array_equal(bits(x), bits(ox)) and np.array_equal(bits(res), bits(ores))
    if ok and np.all(oit >= 0):
        axis = np.array([0.1, -0.2, 1.0]); axis /= np.linalg.norm(axis)
        t, v, f = ComputeForwardVelocitiesFromImu(ctx, gps, rot, acc, axis, batch, shift, iters, 0.01, 3.0, 0.1)
        ot, ov, of = orc.fit_motion_velocities(*gps, *rot, *acc, axis, batch, shift, iters, 0.01, 3.0, 0.1)
        ok = np.array_equal(t, ot) and np.array_equal(bits(v), bits(ov)) and np.array_equal(bits(f), bits(of))
    if not ok:
        raise SystemExit("case %d MISMATCH: n_gps %d, rot %d, acc %d, batch %d, shift %d, iters %d" % (c, len(gps[0]), len(rot[0]), len(acc[0]), batch, shift, iters))
    done += 1
print("fuzz_calib: %d cases equal, %d rejected by both" % (done, skipped))

#!/usr/bin/env python3
"""Generate tests/golden/*.npz.

WHAT THESE VECTORS ARE: the reference (pilotguru / ORB-SLAM2 + OpenCV 2.4.9) cannot be built
or run in the build container (OpenCV, glog, Eigen ... are absent and may not be stubbed), and
its own tests hold no vector for this path.  So these fixtures are
REGRESSION vectors produced by the CPU oracle (oracle/orb_oracle.c) at the commit that first
passed GPU parity -- inputs + expected outputs as data.  They pin the oracle against silent
drift; they do not pin it to the reference ("parity unpinned").

The hand-derivable known answers (Hamming KATs, pattern hash, level sizes, quotas, umax) live
in tests/test_oracle.py itself.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orb_oracle  # noqa: E402
from pilotguru_amd.synth import synth_scene  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CASES = [("a", 0, 320, 240, 500), ("b", 7, 262, 230, 300), ("c", 3, 401, 263, 400)]

for tag, seed, w, h, nf in CASES:
    img = synth_scene(seed, w, h)
    ora = orb_oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    kp, desc = ora.extract(img)
    lv = {}
    for l in range(8):
        c = ora.level_candidates(l)
        lv["cand%d" % l] = np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int16)
        im = ora.level_image(l)
        lv["sum%d" % l] = np.array([int(im.astype(np.uint64).sum()), im.shape[1], im.shape[0]], np.int64)
        lv["nkp%d" % l] = np.array([ora.level_keypoints(l)], np.int32)
    np.savez_compressed(os.path.join(OUT, "extract_%s.npz" % tag), seed=seed, w=w, h=h, nfeatures=nf,
                        kp=kp.view(np.uint8).reshape(len(kp), 28), desc=desc,
                        level7=ora.level_image(7), **lv)
    print(tag, len(kp), os.path.getsize(os.path.join(OUT, "extract_%s.npz" % tag)))

#!/usr/bin/env python3
"""rBRIEF-256 descriptors from scikit-image's ORB descriptor loop (skimage.feature.orb_cy._orb_loop: an
implementation of the published steered-BRIEF test that is independent of OpenCV, of ORB-SLAM2 and of this
repository) on UN-BLURRED images, as a fixture for the oracle's computeOrbDescriptor restatement.

    /opt/conda/bin/python3.9 tests/golden/make_skimage_rbrief.py     # scikit-image 0.18.3 lives only there

What this pins (tests/test_oracle.py::test_rbrief_geometry_and_bit_order_against_scikit_image): the tap
geometry -- row offset x*sin + y*cos, column offset x*cos - y*sin for a pattern point (x, y), i.e.
ORBextractor.cc:118-120 -- the pattern table's row order, and the bit order (test 8 i + m -> bit m of byte i,
LSB first, :123-141).  scikit-image evaluates the taps in double with C round() (half away from zero), the
reference in float with cvRound (half to even): the two agree unless a tap coordinate lies within float
rounding of k + 0.5, and the generator checks that no tap of this fixture does (64 keypoints; it re-seeds otherwise), so
the committed descriptors must be reproduced bit for bit.  Not pinned by this: the blur, sin / cos of the
platform libm, cvRound's tie rule.
"""
import os
import numpy as np
from skimage.feature.orb_cy import _orb_loop
from skimage.feature._orb_descriptor_positions import POS0, POS1


def main():
    factor_pi = np.float32(3.1415926535897932384626433832795 / np.float32(180.0))
    for seed in range(100):
        rng = np.random.RandomState(4000 + seed)
        h, w = 96, 128
        img = np.clip(np.rint(128 + 60 * np.sin(np.arange(w) / 3.0)[None, :] * np.cos(np.arange(h) / 4.0)[:, None]
                              + rng.randint(-40, 41, (h, w))), 0, 255).astype(np.uint8)
        n = 64
        kp = np.stack([rng.randint(20, h - 20, n), rng.randint(20, w - 20, n)], axis=1).astype(np.intp)   # (row, col)
        deg = (rng.uniform(0.0, 360.0, n)).astype(np.float32)
        deg[:4] = [0.25, 359.99, 45.5, 123.456]
        rad32 = (deg * factor_pi).astype(np.float32)               # the reference's float angle (ORBextractor.cc:110)
        # margin check: every rotated tap coordinate at least 5e-6 (several float ulps of a value <= 19) away from a
        # rounding tie, so float-vs-double arithmetic and the tie rule cannot matter
        s, c = np.sin(rad32.astype(np.float64)), np.cos(rad32.astype(np.float64))
        ok = True
        for P in (POS0, POS1):
            pr, pc = P[:, 0].astype(np.float64), P[:, 1].astype(np.float64)
            rr = s[:, None] * pr[None, :] + c[:, None] * pc[None, :]
            cc = c[:, None] * pr[None, :] - s[:, None] * pc[None, :]
            for v in (rr, cc):
                frac = np.abs(v - np.floor(v) - 0.5)
                ok = ok and frac.min() > 5e-6
        if not ok:
            continue
        d = _orb_loop(np.ascontiguousarray(img.astype(np.float64)), np.ascontiguousarray(kp), np.ascontiguousarray(rad32.astype(np.float64)))
        desc = np.packbits(np.asarray(d, dtype=bool), axis=1, bitorder="little")      # test 8i+m -> bit m of byte i
        out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "skimage_rbrief.npz")
        np.savez_compressed(out, img=img, kp_row_col=kp.astype(np.int32), angle_deg=deg, desc=desc,
                            pos=np.concatenate([POS0, POS1], axis=1).astype(np.int8), seed=np.int32(4000 + seed))
        print("wrote", out, "seed", 4000 + seed, desc.shape)
        return
    raise SystemExit("no seed without near-ties found")


if __name__ == "__main__":
    main()

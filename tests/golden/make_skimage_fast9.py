#!/usr/bin/env python3
"""Corner SETS of FAST-9/16 from scikit-image's corner_fast (an implementation of the published
detector that is independent of OpenCV and of this repository), as a fixture for the oracle.

    /opt/conda/bin/python3.9 tests/golden/make_skimage_fast9.py      # scikit-image 0.18.3 there

scikit-image is not in the interpreter the tests run under, so the images and the corner masks
(and orientation angles at random points) are committed as data (tests/golden/skimage_fast9.npz).  What this pins: the corner DECISION
(nine contiguous ring pixels all brighter than p + t or all darker than p - t) of the oracle's
restatement of cv::FAST, pixel for pixel.  What it does not pin: the OpenCV corner score, the
non-max suppression and everything else the oracle restates -- those stay unpinned.

corner_fast works on float images in [0, 1] and compares `ring > p + threshold`; with
threshold = (t + 0.5) / 255 that is `ring >= p + t + 1` for integers, with a margin of 0.5 / 255
against rounding, i.e. exactly OpenCV's strict `ring > p + t`.
"""
import os
import numpy as np
from skimage.feature import corner_fast, corner_orientations
from skimage.feature.orb import OFAST_MASK


def images():
    rng = np.random.RandomState(1234)
    h, w = 120, 160
    yy, xx = np.mgrid[0:h, 0:w]
    # (a) blocks and ramps with a little noise: clean corners, edges, flat areas
    a = np.full((h, w), 60.0)
    for _ in range(40):
        x0, y0 = rng.randint(0, w - 8), rng.randint(0, h - 8)
        a[y0:y0 + rng.randint(4, 30), x0:x0 + rng.randint(4, 40)] = rng.randint(0, 256)
    a += 0.15 * xx + rng.randint(-3, 4, (h, w))
    # (b) pure noise of moderate amplitude: many borderline decisions
    b = 128 + rng.randint(-30, 31, (h, w))
    # (c) smooth blobs (rounded "corners"), low contrast
    c = 120 + 50 * np.sin(xx / 5.0) * np.cos(yy / 7.0) + 20 * np.sin((xx + yy) / 3.0) + rng.randint(-2, 3, (h, w))
    # (d) saturated regions: values at 0 and 255
    d = rng.randint(0, 256, (h, w)).astype(np.float64)
    d[d < 90] = 0
    d[d > 170] = 255
    return [np.clip(np.rint(im), 0, 255).astype(np.uint8) for im in (a, b, c, d)]


def main():
    out = {}
    for i, im in enumerate(images()):
        out["img%d" % i] = im
        for t in (7, 20, 40):
            resp = corner_fast(im, n=9, threshold=(t + 0.5) / 255.0)
            out["mask%d_t%d" % (i, t)] = np.packbits(resp > 0)
    # corner score as "the largest threshold at which the pixel is still a corner" (what OpenCV's
    # cornerScore<16> returns), from scikit-image's DECISION alone: corners are monotone in t, so the
    # score is (number of thresholds 0..254 at which the pixel is a corner) - 1
    for i in (0, 3):
        im = out["img%d" % i]
        cnt = np.zeros(im.shape, np.int32)
        for t in range(0, 255):
            cnt += corner_fast(im, n=9, threshold=(t + 0.5) / 255.0) > 0
        out["maxthr%d" % i] = (cnt - 1).astype(np.int16)        # -1 = never a corner
    # intensity-centroid orientation (Rosin) over the radius-15 disc, the same patch as ORB-SLAM2's
    # IC_Angle: scikit-image's corner_orientations with its ORB mask (749 pixels, the umax table
    # of OpenCV), angle = atan2(m01, m10) in radians, computed in double
    rng = np.random.RandomState(77)
    for i in (0, 1):
        im = out["img%d" % i]
        h, w = im.shape
        pts = np.stack([rng.randint(16, h - 16, 300), rng.randint(16, w - 16, 300)], axis=1)     # (row, col)
        out["orient_pts%d" % i] = pts.astype(np.int32)
        out["orient_rad%d" % i] = corner_orientations(im.astype(np.float64), pts, OFAST_MASK)
    out["ofast_mask"] = OFAST_MASK.astype(np.uint8)
    # scikit-image's copy of the learned rBRIEF pattern (feature/orb_descriptor_positions.txt)
    import skimage
    out["orb_positions"] = np.loadtxt(os.path.join(os.path.dirname(skimage.__file__), "feature",
                                                   "orb_descriptor_positions.txt"), dtype=np.int64).astype(np.int8)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "skimage_fast9.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: int(np.unpackbits(v).sum()) for k, v in out.items() if k.startswith("mask")})


if __name__ == "__main__":
    main()

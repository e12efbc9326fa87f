#!/usr/bin/env python
"""PIN KIT: run this on a machine that has OpenCV 2.4.9 with its Python binding, and send back the two .npz files.

Why.  pilotguru's ORB front end calls five OpenCV primitives -- cv::resize (ORBextractor.cc:1119), cv::FAST (:809-815),
cv::GaussianBlur (:1085), cv::fastAtan2 (:103) and cvRound (:81, :118-120, :436, :1111) -- from OpenCV 2.4.9.1
(docker/Dockerfile:1,21: Ubuntu 14.04's libopencv-dev).  The build container of this repository has no OpenCV in any
form, so the CPU oracle (oracle/orb_oracle.c) RESTATES those primitives from the published 2.4 sources and every parity
claim of the HIP path is "equal to the oracle" -- "parity unpinned" (DESIGN.md section 5).  This script closes that gap
for whoever has the library:

    apt-get install python-opencv            # Ubuntu 14.04 ships 2.4.8/2.4.9; any 2.4.x build of `cv2` will do
    python tools/pin_against_opencv.py       # writes tests/golden/opencv249_primitives.npz and opencv249_extract.npz
    python -m pytest tests/test_opencv_pin.py -q      # (in the repo, with the oracle built) consumes them

It needs numpy and cv2 ONLY (Python 2.7 or 3; no import from this repository), except for the pattern table, which it
reads as DATA from oracle/orb_pattern31.inc (pass --pattern when running outside a checkout).

What it records, on deterministic synthetic scenes generated below (numpy RandomState: identical on every machine):
  primitives   cv2.resize INTER_LINEAR along the reference's pyramid chain (every level of every scene);
               cv2 FAST (FastFeatureDetector, nonmaxSuppression) on whole levels and on 30-px cell windows at
               thresholds 20 and 7 -> (x, y, response) in OpenCV's output order;
               cv2.GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of every level;
               cv2.fastAtan2 on 4096 (y, x) probes (moment-sized integers, axes, octant borders);
               cvRound on half-way and near-half-way doubles (cv2.cv.Round when the build has the legacy module).
  extract      the whole ORBextractor::operator() (ORBextractor.cc:1042-1104) composed FROM THOSE cv2 PRIMITIVES, with
               the in-tree logic -- cell grid and minThFAST retry (:765-829), DistributeOctTree (:539-763), IC_Angle
               (:77-104), computeOrbDescriptor (:107-147) -- restated here in Python line by line: keypoints
               (x, y, size, angle, response, octave) and 256-bit descriptors per scene.

`--backend oracle` runs the very same script against the repository's CPU oracle instead of cv2 (it then imports
oracle.orb_oracle -- the one repository import): that is how the kit itself is tested where there is no OpenCV
(tests/test_opencv_pin.py::test_pin_kit_against_the_oracle_backend), so that a difference reported from a cv2 run is a
difference of the PRIMITIVES, not of this script's restatement.

Known platform dependence: cos / sin at ORBextractor.cc:113 are the platform's libm (here: double cos / sin of the
float angle, rounded once to float -- the oracle's contract); the consuming test reports descriptor
bits that differ and tolerates the handful a 1-ulp cos / sin can flip.
"""
from __future__ import division, print_function

import argparse
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

PATCH_SIZE, HALF_PATCH_SIZE, EDGE_THRESHOLD = 31, 15, 19            # ORBextractor.cc:72-74

# (tag, seed, width, height, nfeatures): small enough for a Python quadtree, odd sizes included
SCENES = [("a", 11, 320, 240, 500), ("b", 12, 401, 263, 400), ("c", 13, 262, 230, 300), ("d", 14, 640, 480, 1000)]


# ---------------------------------------------------------------------------------------------- scenes
def make_scene(seed, w, h):
    """Value noise at three scales + random grey rectangles + +-4 uniform noise (the recipe of SURVEY.md 8d, written
    out with numpy's legacy RandomState so that every machine produces the same bytes)."""
    rng = np.random.RandomState(seed)
    acc = np.zeros((h, w), np.float64)
    for cell, wt in ((64, 4.0), (16, 2.0), (4, 1.0)):
        gh, gw = h // cell + 2, w // cell + 2
        g = rng.randint(0, 256, (gh, gw)).astype(np.float64)
        ys, xs = np.arange(h) / float(cell), np.arange(w) / float(cell)
        y0, x0 = ys.astype(np.int64), xs.astype(np.int64)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        v = (g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0][:, x0 + 1] * (1 - fy) * fx +
             g[y0 + 1][:, x0] * fy * (1 - fx) + g[y0 + 1][:, x0 + 1] * fy * fx)
        acc += wt * v
    img = acc / 7.0
    for _ in range(max(8, w * h // 5000)):
        s = int(rng.randint(8, 65))
        x, y = int(rng.randint(0, w)), int(rng.randint(0, h))
        img[y:y + s, x:x + s] = float(rng.randint(0, 256))
    img += rng.randint(-4, 5, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------- backends
class CvBackend(object):
    def __init__(self):
        import cv2
        self.cv2 = cv2
        self.name = "cv2 " + cv2.__version__
        self._det = {}

    def resize(self, img, w, h):
        return self.cv2.resize(img, (w, h), interpolation=self.cv2.INTER_LINEAR)

    def _detector(self, t):
        if t not in self._det:
            cv2 = self.cv2
            if hasattr(cv2, "FastFeatureDetector_create"):          # (3.x / 4.x API, for cross-version curiosity)
                self._det[t] = cv2.FastFeatureDetector_create(threshold=t, nonmaxSuppression=True)
            else:
                self._det[t] = cv2.FastFeatureDetector(t, True)     # 2.4: FastFeatureDetector(threshold, nonmaxSuppression)
        return self._det[t]

    def fast(self, win, t):
        kps = self._detector(t).detect(np.ascontiguousarray(win), None)
        return [(int(round(k.pt[0])), int(round(k.pt[1])), int(round(k.response))) for k in kps]

    def blur(self, img):
        return self.cv2.GaussianBlur(np.ascontiguousarray(img), (7, 7), 2, None, 2, self.cv2.BORDER_REFLECT_101)

    def atan2(self, y, x):
        return np.float32(self.cv2.fastAtan2(float(y), float(x)))

    def cvround(self, v):
        cv = getattr(self.cv2, "cv", None)
        return int(cv.Round(float(v))) if cv is not None and hasattr(cv, "Round") else None


class OracleBackend(object):
    """The repository's CPU oracle behind the same five calls: the kit's self-test where there is no OpenCV."""

    def __init__(self):
        sys.path.insert(0, ROOT)
        from oracle import orb_oracle
        self.o = orb_oracle
        self.name = "oracle"

    def resize(self, img, w, h):
        return self.o.resize_linear(img, w, h)

    def fast(self, win, t):
        return [(int(c["x"]), int(c["y"]), int(c["response"])) for c in self.o.fast9_nms(np.ascontiguousarray(win), t)]

    def blur(self, img):
        return self.o.gaussian_blur7(np.ascontiguousarray(img), 0)

    def atan2(self, y, x):
        return np.float32(self.o.fast_atan2(np.float32(y), np.float32(x)))

    def cvround(self, v):
        return int(np.rint(np.float64(v)))                           # round half to even == cvtsd2si


# ---------------------------------------------------------------------------------------------- the extractor, line by line
def cv_round(v):
    return int(np.rint(np.float64(v)))                               # cvRound(double) = cvtsd2si: round half to even


def load_pattern(path):
    txt = open(path).read()
    vals = [int(v) for v in txt[txt.index("*/") + 2:].replace("\n", " ").split(",") if v.strip()]
    assert len(vals) == 1024
    return np.array(vals, np.int32).reshape(256, 4)                  # x0 y0 x1 y1 per test (:150-408)


class Extractor(object):
    """ORBextractor (ORBextractor.cc:410-470 constructor; :1042-1104 operator())."""

    def __init__(self, backend, pattern, nfeatures, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        f32 = np.float32
        self.B, self.pattern = backend, pattern
        self.nfeatures, self.nlevels, self.ini_th, self.min_th = nfeatures, nlevels, ini_th, min_th
        sf = [f32(1.0)]
        for _ in range(nlevels):
            sf.append(f32(sf[-1] * f32(scale_factor)))
        self.scale = sf
        self.inv = [f32(f32(1.0) / s) for s in sf]
        factor = f32(f32(1.0) / f32(scale_factor))
        nd = f32(f32(nfeatures) * f32(f32(1) - factor) / f32(f32(1) - f32(math.pow(float(factor), float(nlevels)))))
        self.per_level, total = [], 0
        for _ in range(nlevels):
            self.per_level.append(cv_round(nd)); total += self.per_level[-1]; nd = f32(nd * factor)
        self.per_level.append(max(nfeatures - total, 0))              # (:444-446: the slot behind the last level; unused)
        vmax = int(math.floor(HALF_PATCH_SIZE * math.sqrt(2.0) / 2 + 1))
        vmin = int(math.ceil(HALF_PATCH_SIZE * math.sqrt(2.0) / 2))
        umax = [0] * (HALF_PATCH_SIZE + 2)
        for v in range(vmax + 1):
            umax[v] = cv_round(math.sqrt(HALF_PATCH_SIZE * HALF_PATCH_SIZE - v * v))
        v0 = 0
        for v in range(HALF_PATCH_SIZE, vmin - 1, -1):
            while umax[v0] == umax[v0 + 1]:
                v0 += 1
            umax[v] = v0; v0 += 1
        self.umax = umax[:HALF_PATCH_SIZE + 1]

    def pyramid(self, image):                                         # ComputePyramid (:1106-1131)
        h, w = image.shape
        lv = [image]
        for l in range(1, self.nlevels):
            sw, sh = cv_round(np.float32(np.float32(w) * self.inv[l])), cv_round(np.float32(np.float32(h) * self.inv[l]))
            lv.append(self.B.resize(lv[-1], sw, sh))
        return lv

    def cells(self, img):                                             # the cell loop of ComputeKeyPointsOctTree (:765-829)
        f32 = np.float32
        h, w = img.shape
        minBX = minBY = EDGE_THRESHOLD - 3
        maxBX, maxBY = w - EDGE_THRESHOLD + 3, h - EDGE_THRESHOLD + 3
        width, height = f32(maxBX - minBX), f32(maxBY - minBY)
        nCols, nRows = int(width / f32(30)), int(height / f32(30))
        wCell, hCell = int(math.ceil(width / nCols)), int(math.ceil(height / nRows))
        out = []
        for i in range(nRows):
            iniY = minBY + i * hCell
            maxY = iniY + hCell + 6
            if iniY >= maxBY - 3:
                continue
            maxY = min(maxY, maxBY)
            for j in range(nCols):
                iniX = minBX + j * wCell
                maxX = iniX + wCell + 6
                if iniX >= maxBX - 6:
                    continue
                maxX = min(maxX, maxBX)
                win = img[iniY:maxY, iniX:maxX]
                k = self.B.fast(win, self.ini_th)
                if not k:
                    k = self.B.fast(win, self.min_th)
                out.extend((x + j * wCell, y + i * hCell, r) for x, y, r in k)
        return out, (minBX, maxBX, minBY, maxBY)

    def octtree(self, cand, minX, maxX, minY, maxY, N):               # DistributeOctTree (:539-763)
        f32 = np.float32
        nIni = int(np.floor(f32(maxX - minX) / f32(maxY - minY) + f32(0.5)))      # round() of a positive float
        hX = f32(maxX - minX) / f32(nIni)
        seq = [0]

        def node(ULx, ULy, URx, BRy, keys):
            seq[0] += 1
            return {"b": (ULx, ULy, URx, BRy), "k": keys, "s": seq[0], "nomore": len(keys) == 1}
        roots = [node(int(hX * f32(i)), 0, int(hX * f32(i + 1)), maxY - minY, []) for i in range(nIni)]
        for i, (x, y, r) in enumerate(cand):
            roots[int(f32(x) / hX)]["k"].append(i)
        L = [n for n in roots if n["k"]]
        for n in L:
            n["nomore"] = len(n["k"]) == 1

        def divide(n):                                                # ExtractorNode::DivideNode (:481-537)
            ULx, ULy, URx, BRy = n["b"]
            hx, hy = int(np.ceil(f32(URx - ULx) / 2)), int(np.ceil(f32(BRy - ULy) / 2))
            mx, my = ULx + hx, ULy + hy
            ks = [[], [], [], []]
            for i in n["k"]:
                x, y, _ = cand[i]
                ks[(0 if y < my else 2) if x < mx else (1 if y < my else 3)].append(i)
            bs = [(ULx, ULy, mx, my), (mx, ULy, URx, my), (ULx, my, mx, BRy), (mx, my, URx, BRy)]
            return [node(bs[q][0], bs[q][1], bs[q][2], bs[q][3], ks[q]) for q in range(4)]
        finish = False
        while not finish:
            prev = len(L)
            vec, nexp, front = [], 0, []
            for n in list(L):
                if n["nomore"]:
                    continue
                for ch in divide(n):
                    if ch["k"]:
                        front.insert(0, ch)
                        if len(ch["k"]) > 1:
                            nexp += 1; vec.append(ch)
                L.remove(n)
            L = front + L
            if len(L) >= N or len(L) == prev:
                finish = True
            elif len(L) + 3 * nexp > N:
                while not finish:
                    prev = len(L)
                    # sort by (size, node pointer) at :684; equal sizes: the later-created node first (the parity contract --
                    # the reference's order there depends on the allocator)
                    pv = sorted(vec, key=lambda n: (len(n["k"]), n["s"]))
                    vec = []
                    for n in reversed(pv):
                        for ch in divide(n):
                            if ch["k"]:
                                L.insert(0, ch)
                                if len(ch["k"]) > 1:
                                    vec.append(ch)
                        L.remove(n)
                        if len(L) >= N:
                            break
                    if len(L) >= N or len(L) == prev:
                        finish = True
        out = []
        for n in L:
            best = n["k"][0]
            for i in n["k"][1:]:
                if cand[i][2] > cand[best][2]:
                    best = i
            out.append(best)
        return out

    def ic_angle(self, img, x, y):                                    # IC_Angle (:77-104)
        I = img.astype(np.int64)
        m01 = m10 = 0
        u = np.arange(-HALF_PATCH_SIZE, HALF_PATCH_SIZE + 1)
        m10 += int((u * I[y, x - HALF_PATCH_SIZE:x + HALF_PATCH_SIZE + 1]).sum())
        for v in range(1, HALF_PATCH_SIZE + 1):
            d = self.umax[v]
            uu = np.arange(-d, d + 1)
            plus, minus = I[y + v, x - d:x + d + 1], I[y - v, x - d:x + d + 1]
            m01 += v * int((plus - minus).sum())
            m10 += int((uu * (plus + minus)).sum())
        return self.B.atan2(np.float32(m01), np.float32(m10))

    def descriptor(self, blurred, x, y, angle):                       # computeOrbDescriptor (:107-147)
        f32 = np.float32
        ang = f32(f32(angle) * f32(np.pi / 180.0))                    # factorPI = (float)(CV_PI/180.f)
        a, b = f32(math.cos(float(ang))), f32(math.sin(float(ang)))
        px = self.pattern[:, [0, 2]].astype(np.float32).reshape(-1)   # 512 points: x
        py = self.pattern[:, [1, 3]].astype(np.float32).reshape(-1)
        rr = np.rint(((px * b).astype(np.float32) + (py * a).astype(np.float32)).astype(np.float32).astype(np.float64)).astype(np.int64)
        cc = np.rint(((px * a).astype(np.float32) - (py * b).astype(np.float32)).astype(np.float32).astype(np.float64)).astype(np.int64)
        v = blurred[y + rr, x + cc].astype(np.int32).reshape(256, 2)
        return np.packbits((v[:, 0] < v[:, 1]).astype(np.uint8), bitorder="little") if _has_bitorder() else _packbits_le(v[:, 0] < v[:, 1])

    def __call__(self, image):                                        # operator() (:1042-1104)
        lv = self.pyramid(image)
        kps, descs, stages = [], [], {}
        for l in range(self.nlevels):
            cand, (minBX, maxBX, minBY, maxBY) = self.cells(lv[l])
            stages["cand%d" % l] = np.array(cand, np.int32).reshape(-1, 3)
            sel = self.octtree(cand, minBX, maxBX, minBY, maxBY, self.per_level[l]) if cand else []
            size = np.float32(int(np.float32(PATCH_SIZE) * self.scale[l]))      # `const int scaledPatchSize` (:836)
            level_k = []
            for i in sel:
                x, y, r = cand[i]
                level_k.append([x + minBX, y + minBY, r])
            if not level_k:
                continue
            blurred = self.B.blur(lv[l])
            for x, y, r in level_k:
                ang = self.ic_angle(lv[l], x, y)
                descs.append(self.descriptor(blurred, x, y, ang))
                sx, sy = np.float32(x), np.float32(y)
                if l:
                    sx, sy = np.float32(sx * self.scale[l]), np.float32(sy * self.scale[l])
                kps.append((sx, sy, size, ang, np.float32(r), l))
        kp = np.array(kps, np.float32).reshape(-1, 6)
        return kp, (np.array(descs, np.uint8).reshape(-1, 32) if descs else np.zeros((0, 32), np.uint8)), lv, stages


def _has_bitorder():
    try:
        np.packbits(np.zeros(8, np.uint8), bitorder="little")
        return True
    except TypeError:
        return False


def _packbits_le(bits):
    b = np.asarray(bits, np.uint8).reshape(-1, 8)
    return (b << np.arange(8, dtype=np.uint8)).sum(axis=1).astype(np.uint8)


# ---------------------------------------------------------------------------------------------- probes
def atan2_probes():
    rng = np.random.RandomState(99)
    y = rng.randint(-40000, 40001, 3000).astype(np.float32)
    x = rng.randint(-40000, 40001, 3000).astype(np.float32)
    ax = np.array([0, 1, -1, 5, -5, 12345, -12345, 3, 3, -3, -3, 1000, 1001], np.float32)
    ys = np.concatenate([y, ax, np.zeros_like(ax), ax, -ax, rng.randint(-50, 51, 1032).astype(np.float32)])
    xs = np.concatenate([x, np.zeros_like(ax), ax, ax, ax, rng.randint(-50, 51, 1032).astype(np.float32)])
    return ys[:4096], xs[:4096]


def cvround_probes():
    base = np.arange(-20, 21, dtype=np.float64)
    return np.concatenate([base + 0.5, base + 0.5 - 1e-9, base + 0.5 + 1e-9, base * 2048.0 + 0.5, base / 3.0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=("cv2", "oracle"), default="cv2")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--prefix", default=None, help="file name prefix (default: opencv249 for cv2, oraclepin for the oracle backend)")
    ap.add_argument("--pattern", default=os.path.join(ROOT, "oracle", "orb_pattern31.inc"))
    ap.add_argument("--scenes", default=",".join(s[0] for s in SCENES))
    args = ap.parse_args()
    B = CvBackend() if args.backend == "cv2" else OracleBackend()
    prefix = args.prefix or ("opencv249" if args.backend == "cv2" else "oraclepin")
    pattern = load_pattern(args.pattern)
    prim, ext = {"backend": np.array(B.name)}, {"backend": np.array(B.name)}
    for tag, seed, w, h, nf in SCENES:
        if tag not in args.scenes.split(","):
            continue
        img = make_scene(seed, w, h)
        E = Extractor(B, pattern, nf)
        kp, desc, lv, stages = E(img)
        print("scene %s %dx%d nfeatures %d: %d keypoints (%s)" % (tag, w, h, nf, len(kp), B.name))
        ext["kp_" + tag], ext["desc_" + tag] = kp, desc
        ext["meta_" + tag] = np.array([seed, w, h, nf], np.int64)
        for k, v in stages.items():
            ext[k + "_" + tag] = v
        prim["meta_" + tag] = np.array([seed, w, h, nf], np.int64)
        for l in range(E.nlevels):
            if l:
                prim["resize%d_%s" % (l, tag)] = lv[l]
            if l in (0, 3, 7):
                prim["blur%d_%s" % (l, tag)] = B.blur(lv[l])
        # FAST on a whole level and on a few odd windows, both thresholds, in the detector's output order
        for t in (20, 7):
            prim["fast_level2_t%d_%s" % (t, tag)] = np.array(B.fast(lv[2], t), np.int32).reshape(-1, 3)
            prim["fast_win_t%d_%s" % (t, tag)] = np.array(B.fast(lv[0][20:57, 30:66], t), np.int32).reshape(-1, 3)
            prim["fast_thin_t%d_%s" % (t, tag)] = np.array(B.fast(lv[1][40:47, 10:90], t), np.int32).reshape(-1, 3)
    ys, xs = atan2_probes()
    prim["atan2_y"], prim["atan2_x"] = ys, xs
    prim["atan2"] = np.array([B.atan2(y, x) for y, x in zip(ys, xs)], np.float32)
    pr = cvround_probes()
    got = [B.cvround(v) for v in pr]
    if all(g is not None for g in got):
        prim["cvround_in"], prim["cvround"] = pr, np.array(got, np.int64)
    else:
        print("note: this cv2 build has no cv2.cv.Round; cvRound is pinned through the resize coefficients only")
    if not os.path.isdir(args.out):
        os.makedirs(args.out)
    for name, d in (("primitives", prim), ("extract", ext)):
        path = os.path.join(args.out, "%s_%s.npz" % (prefix, name))
        np.savez_compressed(path, **d)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

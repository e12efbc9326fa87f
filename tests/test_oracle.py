"""CPU tests of the oracle: known answers, committed regression vectors, and independent
numpy restatements of the OpenCV-2.4 stages (so a typo in the C oracle cannot hide).

The reference's own tests hold no vector for this path and the reference
cannot be built here, so the oracle is PARITY-UNPINNED against a real OpenCV 2.4.9 binary."""
import glob
import hashlib
import os

import numpy as np
import pytest
from pilotguru_amd.synth import synth_scene

HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- known answers
def test_pattern_table_hash():
    """sha256 of bit_pattern_31_ (ORBextractor.cc:150-408) as int8[1024]."""
    want = "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    for path in ("oracle/orb_pattern31.inc", "pilotguru_amd/csrc/orb_pattern31.inc"):
        txt = open(os.path.join(HERE, "..", path)).read()
        body = txt[txt.index("*/") + 2:]
        vals = [int(v) for v in body.replace("\n", " ").split(",") if v.strip()]
        assert len(vals) == 1024
        assert vals[:4] == [8, -3, 9, 5] and vals[-4:] == [-1, -6, 0, -11]
        assert max(abs(v) for v in vals) == 13
        assert hashlib.sha256(bytes(v & 0xFF for v in vals)).hexdigest() == want


def test_constructor_tables(oracle):
    """Quotas / umax evaluated from ORBextractor.cc:415-469."""
    o = oracle.OrbOracle(2000, 1.2, 8, 20, 7)
    assert o.features_per_level.tolist() == [434, 362, 302, 251, 209, 175, 145, 121, 1]
    assert oracle.OrbOracle(4000).features_per_level.tolist()[:8] == [869, 724, 603, 503, 419, 349, 291, 242]
    assert oracle.OrbOracle(1000).features_per_level.tolist()[:8] == [217, 181, 151, 126, 105, 87, 73, 61]
    assert o.umax.tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    sf = o.scale_factors
    assert sf[0] == 1.0 and sf[1] == np.float32(1.2)
    acc = np.float32(1.0)
    for i in range(1, 9):                       # float * double -> float  (:421)
        acc = np.float32(np.float64(acc) * np.float64(np.float32(1.2)))
        assert sf[i] == acc
    assert np.array_equal(o.inv_scale_factors, (np.float32(1.0) / sf).astype(np.float32))


def test_level_sizes_1080p_and_480p(oracle):
    """Level sizes quoted in SURVEY.md section 8 (cvRound((float)cols*invScale), :1110-1111)."""
    inv = oracle.OrbOracle(2000).inv_scale_factors
    got = [(int(np.rint(np.float32(1920) * inv[l])), int(np.rint(np.float32(1080) * inv[l]))) for l in range(8)]
    assert got == [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362), (536, 301)]
    assert sum(a * b for a, b in got) == 6419321
    got = [(int(np.rint(np.float32(640) * inv[l])), int(np.rint(np.float32(480) * inv[l]))) for l in range(8)]
    assert sum(a * b for a, b in got) == 950532


def test_descriptor_distance_kats(oracle):
    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0
    assert oracle.descriptor_distance(z, o) == 256
    for bit in (0, 7, 8, 100, 255):
        a = z.copy()
        a[bit // 8] = 1 << (bit % 8)
        assert oracle.descriptor_distance(a, z) == 1
        assert oracle.descriptor_distance(a, o) == 255
    rng = np.random.RandomState(1)
    a = rng.randint(0, 256, (50, 32)).astype(np.uint8)
    b = rng.randint(0, 256, (60, 32)).astype(np.uint8)
    want = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
    assert np.array_equal(oracle.hamming_matrix(a, b), want)
    bi, b1, b2 = oracle.hamming_best2(a, b)
    assert np.array_equal(bi, want.argmin(1))            # argmin = first minimum
    assert np.array_equal(b1, want.min(1))
    assert np.array_equal(b2, np.sort(want, 1)[:, 1])


def test_fast_atan2_known_points(oracle):
    f = oracle.fast_atan2
    assert f(0, 0) == 0.0
    assert abs(f(0, 1) - 0) < 1e-3 and abs(f(1, 0) - 90) < 1e-3
    assert abs(f(0, -1) - 180) < 1e-3 and abs(f(-1, 0) - 270) < 1e-3
    for y, x in ((1, 1), (3, -2), (-5, -7), (-2, 9), (1000, 1), (-1, 1000)):
        true = np.degrees(np.arctan2(y, x)) % 360
        assert abs(f(y, x) - true) < 0.02            # polynomial accuracy of cv::fastAtan2


def test_sincos_contract_close_to_libm(oracle):
    """The parity-contract sin/cos is within 1 float ulp of libm everywhere on [0, 2pi]."""
    for a in np.linspace(0, 2 * np.pi, 4001).astype(np.float32):
        s, c = oracle.sincos_f(a)
        assert abs(np.float32(s) - np.float32(np.sin(np.float64(a)))) <= np.spacing(np.float32(1.0))
        assert abs(np.float32(c) - np.float32(np.cos(np.float64(a)))) <= np.spacing(np.float32(1.0))


def test_rgb_to_gray(oracle):
    rng = np.random.RandomState(0)
    rgb = rng.randint(0, 256, (9, 13, 3)).astype(np.uint8)
    want = ((rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 +
             rgb[..., 2].astype(np.int64) * 1868 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(oracle.rgb_to_gray(rgb), want)


# ---------------------------------------------------------------- independent restatements
def _np_resize(src, dw, dh):
    """cv::resize INTER_LINEAR 8U, vectorised numpy."""
    sh, sw = src.shape

    def coef(d, s):
        scale = 1.0 / (np.float64(d) / s)
        f = ((np.arange(d) + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(np.float32)).astype(np.float32)
        return i, f
    sx, fx = coef(dw, sw)
    fx = np.where((sx < 0) | (sx >= sw - 1), np.float32(0), fx)
    sx = np.clip(sx, 0, sw - 1)
    sx1 = np.minimum(sx + 1, sw - 1)
    a0 = np.rint(((np.float32(1) - fx) * np.float32(2048)).astype(np.float64)).astype(np.int64)
    a1 = np.rint((fx * np.float32(2048)).astype(np.float64)).astype(np.int64)
    sy, fy = coef(dh, sh)
    b0 = np.rint(((np.float32(1) - fy) * np.float32(2048)).astype(np.float64)).astype(np.int64)
    b1 = np.rint((fy * np.float32(2048)).astype(np.float64)).astype(np.int64)
    r0, r1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    S = src.astype(np.int64)
    H0 = S[r0][:, sx] * a0 + S[r0][:, sx1] * a1
    H1 = S[r1][:, sx] * a0 + S[r1][:, sx1] * a1
    out = (((b0[:, None] * (H0 >> 4)) >> 16) + ((b1[:, None] * (H1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 533, 400), (97, 61, 81, 51), (33, 20, 40, 31), (10, 10, 5, 3)])
def test_resize_vs_numpy(oracle, sw, sh, dw, dh):
    rng = np.random.RandomState(sw + dh)
    src = rng.randint(0, 256, (sh, sw)).astype(np.uint8)
    assert np.array_equal(oracle.resize_linear(src, dw, dh), _np_resize(src, dw, dh))


def _np_blur(img, tie_mode):
    t = np.exp(-0.125 * (np.arange(7) - 3.0) ** 2).astype(np.float32)
    k = (t.astype(np.float64) / t.astype(np.float64).sum()).astype(np.float32)
    K = np.rint((k * np.float32(256)).astype(np.float64)).astype(np.int64)
    assert K.tolist() == [18, 34, 49, 55, 49, 34, 18]
    p = np.pad(img.astype(np.int64), 3, mode="reflect")          # numpy 'reflect' == REFLECT_101
    h, w = img.shape
    R = sum(K[i] * p[:, i:i + w] for i in range(7))
    C = sum(K[i] * R[i:i + h, :] for i in range(7))
    v = (C + 32768) >> 16
    if tie_mode == 0:
        tie = ((C & 0xFFFF) == 0x8000) & (np.arange(w)[None, :] < (w & ~3))
        v = np.where(tie, v & ~1, v)
    return np.minimum(v, 255).astype(np.uint8)


@pytest.mark.parametrize("tie", [0, 1])
def test_blur_vs_numpy(oracle, tie):
    rng = np.random.RandomState(3)
    for (h, w) in ((40, 61), (7, 9), (64, 64)):
        img = rng.randint(0, 256, (h, w)).astype(np.uint8)
        assert np.array_equal(oracle.gaussian_blur7(img, tie), _np_blur(img, tie))
    # an exact tie: constant 128 everywhere -> C = 257*257*128 = 8454272 -> 129.0019, no tie;
    # craft one: R=32768/... use brute search for a tie value on a synthetic image
    img = np.zeros((16, 16), np.uint8)
    img[:] = 255
    assert np.all(oracle.gaussian_blur7(img, tie) == 255)          # saturate_cast


_RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2),
         (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _py_fast(img, t):
    """FAST-9/16 + score + NMS straight from the definition (slow, tiny windows only)."""
    h, w = img.shape
    score = np.zeros((h, w), np.int64)
    I = img.astype(np.int64)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = I[y, x]
            ring = [I[y + dy, x + dx] for dx, dy in _RING]
            best = -1
            for sgn in (1, -1):
                d = [sgn * (v - r) for r in ring]
                for k in range(16):
                    m = min(d[(k + j) % 16] for j in range(9))
                    if m > t:
                        best = max(best, m - 1)
            if best >= 0:
                score[y, x] = best
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s and all(s > score[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy):
                out.append((x, y, s))
    return out


def test_fast_vs_definition(oracle):
    from pilotguru_amd.synth import synth_scene
    img = synth_scene(4, 160, 120)
    for (x0, y0, ww, hh, t) in ((10, 10, 37, 37, 20), (60, 40, 36, 31, 7), (100, 70, 41, 20, 7), (0, 0, 7, 7, 7)):
        win = img[y0:y0 + hh, x0:x0 + ww]
        got = oracle.fast9_nms(win, t)
        want = _py_fast(win, t)
        assert [(int(c["x"]), int(c["y"]), int(c["response"])) for c in got] == want
    assert len(oracle.fast9_nms(img[:6, :40], 7)) == 0         # windows under 7 px find nothing


def _py_octtree(cand, minX, maxX, minY, maxY, N):
    """DistributeOctTree (ORBextractor.cc:539-763) with Python lists; tie rule of the parity
    contract: equal sizes -> later created node first."""
    nIni = int(np.floor(np.float32(maxX - minX) / np.float32(maxY - minY) + np.float32(0.5)))
    hX = np.float32(maxX - minX) / np.float32(nIni)
    seq = [0]

    def node(ULx, ULy, URx, BRy, keys):
        seq[0] += 1
        return {"b": (ULx, ULy, URx, BRy), "k": keys, "s": seq[0], "nomore": len(keys) == 1}
    roots = [node(int(hX * np.float32(i)), 0, int(hX * np.float32(i + 1)), maxY - minY, []) for i in range(nIni)]
    for i, (x, y, r) in enumerate(cand):
        roots[int(np.float32(x) / hX)]["k"].append(i)
    L = [n for n in roots if n["k"]]
    for n in L:
        n["nomore"] = len(n["k"]) == 1

    def divide(n):
        ULx, ULy, URx, BRy = n["b"]
        hx = int(np.ceil(np.float32(URx - ULx) / 2))
        hy = int(np.ceil(np.float32(BRy - ULy) / 2))
        mx, my = ULx + hx, ULy + hy
        ks = [[], [], [], []]
        for i in n["k"]:
            x, y, _ = cand[i]
            ks[(0 if y < my else 2) if x < mx else (1 if y < my else 3)].append(i)
        bs = [(ULx, ULy, mx, my), (mx, ULy, URx, my), (ULx, my, mx, BRy), (mx, my, URx, BRy)]
        return [node(*bs[q], ks[q]) for q in range(4)]
    finish = False
    while not finish:
        prev = len(L)
        vec, nexp, newL, i = [], 0, [], 0
        front = []
        for n in list(L):
            if n["nomore"]:
                continue
            for ch in divide(n):
                if ch["k"]:
                    front.insert(0, ch)
                    if len(ch["k"]) > 1:
                        nexp += 1
                        vec.append(ch)
            L.remove(n)
        L = front + L
        if len(L) >= N or len(L) == prev:
            finish = True
        elif len(L) + 3 * nexp > N:
            while not finish:
                prev = len(L)
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


def test_octtree_vs_python_lists(oracle):
    rng = np.random.RandomState(5)
    for (W, H, n, N) in ((288, 208, 900, 100), (600, 200, 3000, 217), (100, 100, 40, 60), (300, 120, 5, 30),
                         (288, 208, 2500, 61), (500, 90, 700, 120)):
        pts = set()
        while len(pts) < n:
            pts.add((int(rng.randint(3, W - 3)), int(rng.randint(3, H - 3))))
        pts = sorted(pts, key=lambda p: (p[1], p[0]))
        cand = np.zeros(n, oracle.CAND_DTYPE)
        cand["x"] = [p[0] for p in pts]
        cand["y"] = [p[1] for p in pts]
        cand["response"] = rng.randint(7, 40, n)              # many response ties
        got = oracle.distribute_octtree(cand, 16, 16 + W, 16, 16 + H, N)
        want = _py_octtree([(int(c["x"]), int(c["y"]), int(c["response"])) for c in cand], 16, 16 + W, 16, 16 + H, N)
        assert got.tolist() == want


def test_descriptor_vs_numpy(oracle):
    from pilotguru_amd.synth import synth_scene
    img = synth_scene(2, 96, 96)
    txt = open(os.path.join(HERE, "..", "oracle", "orb_pattern31.inc")).read()
    pat = np.array([int(v) for v in txt[txt.index("*/") + 2:].replace("\n", " ").split(",") if v.strip()]).reshape(256, 4)
    for angle in (0.0, 33.3, 90.0, 181.25, 359.9):
        s, c = oracle.sincos_f(np.float32(angle) * np.float32(np.pi / 180.0))
        a, b = np.float32(c), np.float32(s)
        bits = []
        for x0, y0, x1, y1 in pat:
            def tap(px, py):
                r = int(np.rint(np.float64(np.float32(np.float32(px) * b) + np.float32(np.float32(py) * a))))
                cc = int(np.rint(np.float64(np.float32(np.float32(px) * a) - np.float32(np.float32(py) * b))))
                return int(img[48 + r, 48 + cc])
            bits.append(1 if tap(x0, y0) < tap(x1, y1) else 0)
        want = np.packbits(np.array(bits, np.uint8), bitorder="little")
        assert np.array_equal(oracle.orb_descriptor(img, 48, 48, angle), want)


def test_ic_angle_vs_numpy(oracle):
    from pilotguru_amd.synth import synth_scene
    img = synth_scene(6, 80, 80)
    umax = oracle.OrbOracle(500).umax
    m10 = m01 = 0
    for v in range(-15, 16):
        for u in range(-int(umax[abs(v)]), int(umax[abs(v)]) + 1):
            m10 += u * int(img[40 + v, 40 + u])
            m01 += v * int(img[40 + v, 40 + u])
    assert oracle.ic_angle(img, 40, 40, umax) == oracle.fast_atan2(np.float32(m01), np.float32(m10))


# ---------------------------------------------------------------- committed regression vectors
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "extract_*.npz"))))
def test_oracle_matches_committed_vectors(oracle, path):
    from pilotguru_amd.synth import synth_scene
    g = np.load(path)
    img = synth_scene(int(g["seed"]), int(g["w"]), int(g["h"]))
    ora = oracle.OrbOracle(int(g["nfeatures"]), 1.2, 8, 20, 7)
    kp, desc = ora.extract(img)
    assert kp.view(np.uint8).reshape(len(kp), 28).tobytes() == g["kp"].tobytes()
    assert np.array_equal(desc, g["desc"])
    assert np.array_equal(ora.level_image(7), g["level7"])
    for l in range(8):
        c = ora.level_candidates(l)
        assert np.array_equal(np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int16), g["cand%d" % l])
        im = ora.level_image(l)
        assert [int(im.astype(np.uint64).sum()), im.shape[1], im.shape[0]] == g["sum%d" % l].tolist()
        assert ora.level_keypoints(l) == int(g["nkp%d" % l][0])


def test_quota_overshoot_and_order_properties(oracle):
    """A level never returns more than max(quota + 2, 4*nIni) keypoints; octaves are emitted in
    level order; every keypoint is >= 19 px from the level border (16-px region + FAST margin)."""
    from pilotguru_amd.synth import synth_scene
    img = synth_scene(9, 480, 360)
    ora = oracle.OrbOracle(800, 1.2, 8, 20, 7)
    kp, desc = ora.extract(img)
    q = ora.features_per_level
    assert np.all(np.diff(kp["octave"]) >= 0)
    sf = ora.scale_factors
    for l in range(8):
        n = ora.level_keypoints(l)
        assert n <= max(q[l] + 2, 8)
        w, h = ora.level_size(l)
        k = kp[kp["octave"] == l]
        x = np.rint(k["x"] / sf[l])
        y = np.rint(k["y"] / sf[l])
        assert np.all(x >= 19) and np.all(x < w - 19) and np.all(y >= 19) and np.all(y < h - 19)
    assert np.all(kp["class_id"] == -1)


def test_fast9_corner_set_matches_scikit_image(oracle):
    """The oracle's FAST-9/16 corner DECISION against scikit-image's corner_fast, an
    implementation independent of OpenCV and of this repository (fixture + generator:
    tests/golden/skimage_fast9.npz, make_skimage_fast9.py; scikit-image is not installed in the
    test interpreter).  Pins the corner set pixel for pixel on four images x three thresholds;
    the OpenCV score / NMS restatements stay unpinned."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_fast9.npz"))
    checked = 0
    for i in range(4):
        img = fx["img%d" % i]
        h, w = img.shape
        for t in (7, 20, 40):
            want = np.unpackbits(fx["mask%d_t%d" % (i, t)])[: h * w].reshape(h, w).astype(bool)
            got = oracle.fast9_score_map(img, t) > 0
            # scikit-image leaves a 3-px border untested, like cv::FAST
            assert not want[:3].any() and not want[-3:].any() and not want[:, :3].any() and not want[:, -3:].any()
            assert np.array_equal(got, want), "image %d threshold %d: %d pixels differ" % (i, t, int((got != want).sum()))
            checked += int(want.sum())
    assert checked > 10000


def test_ic_angle_matches_scikit_image_orientation(oracle):
    """IC_Angle (integer moments over the radius-15 disc + cv::fastAtan2) against scikit-image's
    corner_orientations on its ORB mask (double-precision atan2): same disc (749 pixels), same
    axis convention, angles equal up to the 0.3 degree error of fastAtan2's polynomial."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_fast9.npz"))
    ora = oracle.OrbOracle(500, 1.2, 8, 20, 7)
    umax = ora.umax
    # the disc of the umax table is scikit-image's OFAST mask
    mask = np.zeros((31, 31), np.uint8)
    for v in range(-15, 16):
        u = umax[abs(v)]
        mask[v + 15, 15 - u:15 + u + 1] = 1
    assert np.array_equal(mask, fx["ofast_mask"]) and int(mask.sum()) == 749
    worst = 0.0
    for i in (0, 1):
        img = fx["img%d" % i]
        for (r, c), rad in zip(fx["orient_pts%d" % i], fx["orient_rad%d" % i]):
            deg = oracle.ic_angle(img, int(c), int(r), umax)
            want = np.degrees(rad) % 360.0
            diff = abs((deg - want + 180.0) % 360.0 - 180.0)
            worst = max(worst, diff)
    assert worst < 0.35, worst


def test_resize_geometry_matches_torch_bilinear(oracle):
    """cv::resize(INTER_LINEAR) restatement against torch's bilinear interpolation (half-pixel
    centres, no antialiasing: the same sampling geometry, but float arithmetic instead of OpenCV's
    11-bit fixed point): every pixel within one grey level, most identical after rounding, bias below 0.2.
    A library-independent check of the geometry / border convention, not of the bit-exact
    fixed-point arithmetic (that is what the numpy restatement above covers)."""
    import torch
    rng = np.random.RandomState(5)
    for (w, h, dw, dh) in [(320, 240, 267, 200), (641, 479, 534, 399), (100, 80, 50, 40), (97, 131, 81, 109)]:
        src = (rng.randint(0, 256, (h, w)) * 0.5 + 64 + 40 * np.sin(np.arange(w) / 9.0)[None, :]).clip(0, 255).astype(np.uint8)
        got = oracle.resize_linear(src, dw, dh).astype(np.int32)
        t = torch.from_numpy(src.astype(np.float64))[None, None]
        ref = torch.nn.functional.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0, 0].numpy()
        assert np.abs(got - ref).max() <= 1.0 + 1e-9, (w, h, np.abs(got - ref).max())
        assert (got == np.rint(ref).astype(np.int32)).mean() > 0.8
        assert abs((got - ref).mean()) < 0.2      # the truncating fixed-point path sits ~0.13 grey levels low


def test_gaussian_blur_matches_scipy_float_convolution(oracle):
    """cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) restatement (8-bit fixed-point kernel,
    two-stage rounding) against scipy's float separable correlation with the rounded Gaussian taps and
    mode='mirror' (the same border rule): within one grey level everywhere, borders included."""
    from scipy import ndimage
    rng = np.random.RandomState(11)
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / (2.0 * 2.0 * 2.0)); k /= k.sum()
    k = np.rint(k * 256.0) / 256.0                       # OpenCV's 8-bit taps 18 34 49 55 49 34 18 (sum 257/256)
    assert np.array_equal(k * 256, [18, 34, 49, 55, 49, 34, 18])
    for (w, h) in [(64, 48), (131, 97), (40, 200)]:
        img = (rng.randint(0, 256, (h, w)) * 0.6 + 50 + 30 * np.cos(np.arange(h) / 6.0)[:, None]).clip(0, 255).astype(np.uint8)
        got = oracle.gaussian_blur7(img).astype(np.float64)
        ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        assert np.abs(got - ref).max() < 1.0, np.abs(got - ref).max()
        assert abs((got - ref).mean()) < 0.1


def test_fast9_corner_score_is_largest_corner_threshold(oracle):
    """cv::FAST's corner score (cornerScore<16>: best arc minimum - 1) is the largest threshold at
    which the pixel is still a FAST-9 corner.  The fixture holds that quantity computed from
    scikit-image's corner DECISION at every threshold 0..254 -- so this pins the oracle's score
    values, not only its corner set, against an independent implementation."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_fast9.npz"))
    n = 0
    for i in (0, 3):
        img, maxthr = fx["img%d" % i], fx["maxthr%d" % i].astype(np.int32)
        for t in (7, 20, 40):
            sc = oracle.fast9_score_map(img, t).astype(np.int32)
            corner = maxthr >= t
            assert np.array_equal(sc > 0, corner)
            assert np.array_equal(sc[corner], maxthr[corner])
            n += int(corner.sum())
    assert n > 3000


def test_pattern_table_equals_scikit_image_copy():
    """The 256 x (x0, y0, x1, y1) rBRIEF test pairs equal scikit-image's copy of the learned ORB
    pattern (feature/orb_descriptor_positions.txt in scikit-image 0.18.3), number for number."""
    fx = np.load(os.path.join(HERE, "golden", "skimage_fast9.npz"))
    txt = open(os.path.join(HERE, "..", "pilotguru_amd", "csrc", "orb_pattern31.inc")).read()
    body = txt[txt.index("*/") + 2:]
    vals = np.array([int(v) for v in body.replace("\n", " ").split(",") if v.strip()], np.int8).reshape(256, 4)
    assert np.array_equal(vals, fx["orb_positions"])


def test_ingest_geometry_is_rot90_and_flips(oracle):
    """The reader's rotation (cv::flip of the transpose, image_sequence_reader.cc:186-205) and
    flips (:53-58) restated literally equal numpy's rot90 (counter-clockwise quarter turns) and
    slicing, for grey and colour frames and all 16 combinations; other angles are rejected."""
    rng = np.random.RandomState(3)
    for shape in [(5, 7), (9, 4, 3), (6, 6, 4)]:
        img = rng.randint(0, 256, shape).astype(np.uint8)
        for rot, k in ((0, 0), (90, 1), (180, 2), (270, 3)):
            for vf in (False, True):
                for hf in (False, True):
                    ref = np.rot90(img, k)
                    if vf: ref = ref[::-1]
                    if hf: ref = ref[:, ::-1]
                    assert np.array_equal(oracle.ingest_geometry(img, rot, vf, hf), ref), (shape, rot, vf, hf)
    with pytest.raises(ValueError):
        oracle.ingest_geometry(rng.randint(0, 256, (4, 4)).astype(np.uint8), 45)


def test_rbrief_geometry_and_bit_order_against_scikit_image(oracle):
    """computeOrbDescriptor (ORBextractor.cc:107-147) against scikit-image's independent steered-BRIEF loop on an
    un-blurred image (fixture + generator: tests/golden/make_skimage_rbrief.py): tap geometry (row = x sin + y cos,
    column = x cos - y sin), pattern row order and LSB-first bit packing -- all 64 x 256 bits equal; the fixture
    avoids rounding near-ties, so float vs double and the tie rule cannot matter.  scikit-image ships its own copy
    of the 256 x 4 pattern: it must equal the table the oracle and the kernels use."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_rbrief.npz"))
    img, kp, deg, want = fx["img"], fx["kp_row_col"], fx["angle_deg"], fx["desc"]
    txt = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "orb_pattern31.inc")).read()
    body = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith(("/", "*")))
    nums = np.array([int(t) for t in body.replace(",", " ").split() if t.lstrip("-").isdigit()], np.int64)
    assert nums.size == 1024 and np.array_equal(nums.reshape(256, 4), fx["pos"].astype(np.int64))
    got = np.stack([oracle.orb_descriptor(img, int(c), int(r), float(a)) for (r, c), a in zip(kp, deg)])
    assert got.shape == want.shape == (64, 32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("tie", [0, 1])
def test_gaussian_blur_integer_arithmetic_against_scipy_int32(oracle, tie):
    """The blur's INTEGER arithmetic against scipy.ndimage.correlate1d run on int32 with the
    exact Q8 taps 18 34 49 55 49 34 18 and mode='mirror' (= BORDER_REFLECT_101): row sums, column sums and the
    final (C + 32768) >> 16 with the stated tie rule -- every pixel EQUAL, not "within one grey level"."""
    from scipy import ndimage
    K = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    rng = np.random.RandomState(21 + tie)
    for (w, h) in [(64, 48), (131, 97), (40, 200), (7, 9)]:
        img = rng.randint(0, 256, (h, w)).astype(np.uint8)
        img[: h // 3] = (img[: h // 3] // 64) * 64                     # plateaus: more exact .5 ties
        R = ndimage.correlate1d(img.astype(np.int64), K, axis=1, mode="mirror")
        Cc = ndimage.correlate1d(R, K, axis=0, mode="mirror")
        v = (Cc + 32768) >> 16
        if tie == 0:                                                    # SSE2 column pass: ties to even for x < (w & ~3)
            is_tie = (Cc & 0xFFFF) == 0x8000
            xs = np.arange(w)[None, :] < (w & ~3)
            v = np.where(is_tie & xs, v & ~1, v)
        want = np.clip(v, 0, 255).astype(np.uint8)
        got = oracle.gaussian_blur7(img, tie_mode=tie)
        assert np.array_equal(got, want), (w, h)


# ---- the SIMD variants of what OpenCV 2.4.9 vectorises (oracle/orb_simd.c): bit-equal to the scalar restatement ----------------

def _simd_or_skip(oracle):
    if not oracle.simd_available():
        pytest.skip("this CPU has no AVX2: the SIMD variants of the oracle do not run here")


@pytest.mark.parametrize("w,h", [(37, 37), (36, 31), (7, 7), (8, 40), (19, 22), (23, 9), (64, 48), (131, 67), (6, 30)])
def test_simd_fast_equals_scalar(oracle, w, h):
    """cv::FAST's vector form (16 pixels per vector: compass pre-test, 25-step run count on bytes, cornerScore on eight 16-bit
    lanes; scalar row tail) against the scalar oracle: the score of EVERY pixel and the NMS output, on noise (dense corners, equal
    neighbouring scores), on texture, on flat images with saturating v +- t, at both thresholds of the path (20, 7) and the extremes."""
    _simd_or_skip(oracle)
    rng = np.random.RandomState(w * 100 + h)
    imgs = [rng.randint(0, 256, (h, w)).astype(np.uint8), (128 + rng.randint(-12, 13, (h, w))).astype(np.uint8),
            np.full((h, w), 250, np.uint8), (rng.randint(0, 2, (h, w)) * 255).astype(np.uint8), synth_scene(w + h, max(w, 64), max(h, 64))[:h, :w]]
    for img in imgs:
        for t in (20, 7, 1, 60, 200, 255):
            assert np.array_equal(oracle.fast9_score_map(img, t, simd=True), oracle.fast9_score_map(img, t)), (w, h, t)
            a, b = oracle.fast9_nms(img, t, simd=True), oracle.fast9_nms(img, t)
            assert a.tobytes() == b.tobytes(), (w, h, t)


@pytest.mark.parametrize("sw,sh", [(1920, 1080), (641, 479), (333, 251), (100, 150), (17, 9), (9, 8)])
def test_simd_resize_and_blur_equal_scalar(oracle, sw, sh):
    """cv::resize's vertical pass (8 x 32-bit lanes) and both GaussianBlur passes, in both tie modes, against the scalar oracle
    at the pyramid's scale factors and at odd sizes (vector tails of every length, images narrower than a vector)."""
    _simd_or_skip(oracle)
    rng = np.random.RandomState(sw + sh)
    img = rng.randint(0, 256, (sh, sw)).astype(np.uint8)
    for f in (1.2, 1.5, 2.0, 1.07):
        dw, dh = max(int(round(sw / f)), 1), max(int(round(sh / f)), 1)
        assert np.array_equal(oracle.resize_linear(img, dw, dh, simd=True), oracle.resize_linear(img, dw, dh)), (sw, sh, f)
    flat = np.full((sh, sw), 77, np.uint8); flat[::2, ::3] = 78                 # many exact ties in the column pass
    for im in (img, flat, synth_scene(3, max(sw, 64), max(sh, 64))[:sh, :sw]):
        for tie in (0, 1):
            assert np.array_equal(oracle.gaussian_blur7(im, tie, simd=True), oracle.gaussian_blur7(im, tie)), (sw, sh, tie)


@pytest.mark.parametrize("w,h,nf,scale,nlev", [(640, 480, 1000, 1.2, 8), (1280, 720, 2000, 1.2, 8), (333, 251, 500, 1.5, 3)])
def test_simd_extract_equals_scalar_extract(oracle, w, h, nf, scale, nlev):
    """The whole extractor with the SIMD primitives switched in (bench.py's "port+simd" CPU baseline): every pyramid level, every
    candidate, every keypoint and descriptor equal to the scalar oracle's -- on a textured, a driving-like and a noise frame."""
    _simd_or_skip(oracle)
    from pilotguru_amd.synth import synth_scene_road
    rng = np.random.RandomState(5)
    for img in (synth_scene(11, w, h), synth_scene_road(4, w, h), (128 + rng.randint(-9, 10, (h, w))).astype(np.uint8)):
        a, b = oracle.OrbOracle(nf, scale, nlev, 20, 7, simd=True), oracle.OrbOracle(nf, scale, nlev, 20, 7)
        assert a.simd and not b.simd
        ka, da = a.extract(img); kb, db = b.extract(img)
        assert ka.tobytes() == kb.tobytes() and np.array_equal(da, db)
        for l in range(nlev):
            assert np.array_equal(a.level_image(l), b.level_image(l))
            assert a.level_candidates(l).tobytes() == b.level_candidates(l).tobytes()

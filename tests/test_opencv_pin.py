"""The OpenCV 2.4.9 pin.

The oracle's cv::resize / cv::FAST / cv::GaussianBlur / cv::fastAtan2 / cvRound are restated from the published OpenCV 2.4
sources: this image has no OpenCV, so they cannot be pinned here.  tools/pin_against_opencv.py is the kit for whoever has
the library (docker/Dockerfile:1,21 of the reference: Ubuntu 14.04's 2.4.9.1): it records the five primitives and a whole
ORBextractor::operator() composed from them on deterministic scenes into tests/golden/opencv249_*.npz.

  * test_pin_kit_against_the_oracle_backend   runs the kit with `--backend oracle` (the same script, the oracle's primitives
    behind the five calls) and holds its Python restatement of the in-tree logic -- cell loop + minThFAST retry, quadtree,
    IC_Angle, rBRIEF, output assembly -- to the oracle's C pipeline, keypoint bytes and descriptors.  So when a cv2 run
    disagrees with the oracle, the disagreement is in a PRIMITIVE, and the per-primitive records say which.
  * test_oracle_against_opencv249            consumes tests/golden/opencv249_*.npz when someone has produced them; until
    then it SKIPS, loudly, with the command to run -- that is the "parity unpinned" of DESIGN.md, as a test outcome.
"""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KIT = os.path.join(ROOT, "tools", "pin_against_opencv.py")


def _kit():
    spec = importlib.util.spec_from_file_location("pin_against_opencv", KIT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _compare(oracle, prim, ext, exact_descriptors):
    """Every record of a (primitives, extract) pair of files against the oracle.  Returns a report dict; asserts on
    anything but the descriptor bits a 1-ulp cos / sin may flip (counted, bounded)."""
    kit = _kit()
    report = {"backend": str(prim["backend"]), "scenes": 0, "resize_levels": 0, "blur_images": 0, "fast_sets": 0, "descriptor_bits_differing": 0,
              "keypoints": 0}
    tags = sorted(k[5:] for k in prim.files if k.startswith("meta_"))
    for tag in tags:
        seed, w, h, nf = [int(v) for v in prim["meta_" + tag]]
        img = kit.make_scene(seed, w, h)
        report["scenes"] += 1
        ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
        # cv::resize, step by step: level l from the RECORDED level l - 1 (an error does not propagate into the next check)
        prev = img
        for l in range(1, 8):
            want = prim["resize%d_%s" % (l, tag)]
            got = oracle.resize_linear(prev, want.shape[1], want.shape[0])
            assert (want.shape[1], want.shape[0]) == ora_level_size(oracle, w, h, l), "level size (cvRound) of level %d, scene %s" % (l, tag)
            assert np.array_equal(got, want), "cv::resize INTER_LINEAR, scene %s level %d: %d pixels differ" % (tag, l, int((got != want).sum()))
            prev = want
            report["resize_levels"] += 1
        levels = [img] + [prim["resize%d_%s" % (l, tag)] for l in range(1, 8)]
        # cv::GaussianBlur 7x7 sigma 2 REFLECT_101 (tie mode 0 = the SSE2 build's rule, DESIGN.md section 5 contract 4)
        for l in (0, 3, 7):
            want = prim["blur%d_%s" % (l, tag)]
            got = oracle.gaussian_blur7(levels[l], 0)
            if not np.array_equal(got, want):
                alt = oracle.gaussian_blur7(levels[l], 1)
                assert np.array_equal(alt, want), "cv::GaussianBlur, scene %s level %d: %d pixels differ under either tie rule" % (
                    tag, l, int((got != want).sum()))
                report.setdefault("blur_tie_mode_1_needed", []).append((tag, l))
            report["blur_images"] += 1
        # cv::FAST with nonmaxSuppression: the same corners, the same responses, the same ORDER
        for t in (20, 7):
            for name, win in (("fast_level2", levels[2]), ("fast_win", levels[0][20:57, 30:66]), ("fast_thin", levels[1][40:47, 10:90])):
                want = prim["%s_t%d_%s" % (name, t, tag)]
                got = oracle.fast9_nms(np.ascontiguousarray(win), t)
                got = np.stack([got["x"], got["y"], got["response"]], 1).astype(np.int32).reshape(-1, 3)
                assert np.array_equal(got, want), "cv::FAST %s t=%d scene %s: %d vs %d corners" % (name, t, tag, len(got), len(want))
                report["fast_sets"] += 1
        # the whole extractor
        kp, desc = ora.extract(img)
        wk, wd = ext["kp_" + tag], ext["desc_" + tag]
        for l in range(8):
            c = ora.level_candidates(l)
            got = np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32).reshape(-1, 3)
            assert np.array_equal(got, ext["cand%d_%s" % (l, tag)]), "vToDistributeKeys of level %d, scene %s" % (l, tag)
        assert len(kp) == len(wk), "scene %s: %d keypoints, recorded %d" % (tag, len(kp), len(wk))
        mine = np.stack([kp["x"], kp["y"], kp["size"], kp["angle"], kp["response"], kp["octave"].astype(np.float32)], 1)
        assert np.array_equal(mine.view(np.uint32), wk.view(np.uint32)), "scene %s: keypoint records (x, y, size, angle, response, octave)" % tag
        bits = int(np.unpackbits(desc ^ wd).sum())
        report["descriptor_bits_differing"] += bits
        report["keypoints"] += len(kp)
        if exact_descriptors:
            assert bits == 0, "scene %s: %d descriptor bits differ" % (tag, bits)
    # cv::fastAtan2
    got = np.array([oracle.fast_atan2(y, x) for y, x in zip(prim["atan2_y"], prim["atan2_x"])], np.float32)
    assert np.array_equal(got.view(np.uint32), prim["atan2"].view(np.uint32)), "cv::fastAtan2: %d of %d probes differ" % (
        int((got.view(np.uint32) != prim["atan2"].view(np.uint32)).sum()), len(got))
    if "cvround" in prim.files:
        assert np.array_equal(np.rint(prim["cvround_in"]).astype(np.int64), prim["cvround"]), "cvRound is not round-half-to-even"
        report["cvround_probes"] = len(prim["cvround"])
    # cos / sin at ORBextractor.cc:113 are the platform's: a 1-ulp difference moves a tap of ~1e-7 of the keypoints
    assert report["descriptor_bits_differing"] <= max(2, report["keypoints"] // 200), report
    return report


def ora_level_size(oracle, w, h, l):
    o = oracle.OrbOracle(100, 1.2, 8, 20, 7)
    inv = o.inv_scale_factors
    return (int(np.rint(np.float64(np.float32(np.float32(w) * inv[l])))), int(np.rint(np.float64(np.float32(np.float32(h) * inv[l])))))


def test_pin_kit_against_the_oracle_backend(oracle, tmp_path):
    r = subprocess.run([sys.executable, KIT, "--backend", "oracle", "--out", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True)
    assert r.returncode == 0, r.stdout
    prim = np.load(os.path.join(str(tmp_path), "oraclepin_primitives.npz"))
    ext = np.load(os.path.join(str(tmp_path), "oraclepin_extract.npz"))
    rep = _compare(oracle, prim, ext, exact_descriptors=False)
    assert rep["scenes"] == 4 and rep["resize_levels"] == 28 and rep["fast_sets"] == 24 and rep["keypoints"] > 2000
    # the kit's cos / sin (double, rounded once) against the oracle's contract on ~2 200 keypoints: no bit may differ
    assert rep["descriptor_bits_differing"] == 0, rep


def test_oracle_against_opencv249(oracle):
    p1, p2 = os.path.join(HERE, "golden", "opencv249_primitives.npz"), os.path.join(HERE, "golden", "opencv249_extract.npz")
    if not (os.path.exists(p1) and os.path.exists(p2)):
        pytest.skip("PARITY UNPINNED: tests/golden/opencv249_*.npz are absent -- nobody has run `python tools/pin_against_opencv.py` "
                    "on a machine with OpenCV 2.4.9 yet (this image has no OpenCV; DESIGN.md section 5).  The oracle's resize / FAST / "
                    "GaussianBlur / fastAtan2 / cvRound are restated from the published sources, not checked against the library.")
    rep = _compare(oracle, np.load(p1), np.load(p2), exact_descriptors=False)
    print("OpenCV pin:", rep)

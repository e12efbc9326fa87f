"""ctypes loader for the CPU oracle (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (pilotguru_amd) never does.  See
oracle/orb_oracle.h for the "PARITY UNPINNED" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGORB_ORACLE_LIB: another build of the same sources (bench.py's cpu_baseline leg times a -march=native one)
_LIB = os.environ.get("PGORB_ORACLE_LIB") or os.path.join(_HERE, "liborb_oracle.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<i4")])


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in ("orb_oracle.c", "orb_oracle.h", "orb_pattern31.inc", "post_oracle.c", "calib_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborb_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        u8p, i32p, u16p, fp = (C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                               C.POINTER(C.c_uint16), C.POINTER(C.c_float))
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        for name in ("orc_scale_factors", "orc_inv_scale_factors", "orc_level_sigma2",
                     "orc_inv_level_sigma2"):
            getattr(L, name).restype = fp
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("orc_features_per_level", "orc_umax"):
            getattr(L, name).restype = i32p
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_extract.restype = C.c_int
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, i32p]
        L.orc_level_size.argtypes = [C.c_void_p, C.c_int, i32p, i32p]
        L.orc_level_image.restype = C.c_void_p
        L.orc_level_image.argtypes = [C.c_void_p, C.c_int]
        L.orc_level_blurred.restype = C.c_void_p
        L.orc_level_blurred.argtypes = [C.c_void_p, C.c_int]
        L.orc_level_candidates.restype = C.c_int
        L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.orc_level_keypoints.restype = C.c_int
        L.orc_level_keypoints.argtypes = [C.c_void_p, C.c_int]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_fast9_nms.restype = C.c_int
        L.orc_fast9_nms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_ingest_geometry.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_fast9_score_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_fast9_score_map.restype = None
        L.orc_distribute_octtree.restype = C.c_int
        L.orc_distribute_octtree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, i32p]
        L.orc_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_sincos_f.argtypes = [C.c_float, fp, fp]
        L.orc_sincos_checksum.restype = C.c_uint64
        L.orc_sincos_checksum.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_orb_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orc_rgb_to_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        # SIMD variants of the three primitives OpenCV 2.4.9 vectorises (orb_simd.c): bit-equal to the scalar functions
        L.orc_simd_available.restype = C.c_int
        L.orc_set_simd.restype = C.c_int
        L.orc_set_simd.argtypes = [C.c_void_p, C.c_int]
        L.orc_resize_linear_u8_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_gaussian_blur7_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_fast9_score_map_simd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_fast9_score_map_simd.restype = None
        L.orc_fast9_nms_simd.restype = C.c_int
        L.orc_fast9_nms_simd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_descriptor_distance.restype = C.c_int
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_hamming_matrix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_hamming_best2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_frame_grid.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_features_in_area.restype = C.c_int
        L.orc_features_in_area.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_float] * 7 + [C.c_int, C.c_int, C.c_void_p]
        L.orc_search_for_initialization.restype = C.c_int
        L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                                    C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int]
        L.orc_search_by_projection_points.restype = C.c_int
        L.orc_search_by_projection_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + \
            [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_float, C.c_float, C.c_void_p]
        L.orc_search_by_projection_frame.restype = C.c_int
        L.orc_search_by_projection_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + \
            [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_float, C.c_int, C.c_void_p]
        L.orc_search_by_projection_keyframe.restype = C.c_int
        L.orc_search_by_projection_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + \
            [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int] + [C.c_void_p] * 9 + [C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.orc_log_f.restype = C.c_float
        L.orc_log_f.argtypes = [C.c_float]
        L.orc_predict_scale.restype = C.c_int
        L.orc_predict_scale.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
        L.orc_search_by_bow.restype = C.c_int
        L.orc_search_by_bow.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int] + \
            [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_undistort_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_image_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_vocab_load_text.restype = C.c_void_p
        L.orc_vocab_load_text.argtypes = [C.c_char_p]
        L.orc_vocab_free.argtypes = [C.c_void_p]
        L.orc_vocab_info.argtypes = [C.c_void_p, i32p, i32p, i32p, i32p]
        L.orc_bow_transform_one.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
        L.orc_bow_score_l1.restype = C.c_double
        L.orc_bow_score_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrbOracle:
    """Mirror of ORB_SLAM2::ORBextractor (ref: include/ORBextractor.h:44-110)."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th_fast=20,
                 min_th_fast=7, blur_tie_mode=0, simd=False):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orc_create(nfeatures, scale_factor, nlevels, ini_th_fast,
                                   min_th_fast, blur_tie_mode)
        if not self.h:
            raise ValueError("orc_create failed")
        # simd: FAST, the vertical resize pass and both blur passes through their SIMD variants (oracle/orb_simd.c: what OpenCV
        # 2.4.9 vectorises; same results bit for bit).  .simd says whether this CPU took them (AVX2).
        self.simd = bool(self.L.orc_set_simd(self.h, 1 if simd else 0))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def _tab(self, fn, n, dt):
        return np.ctypeslib.as_array(fn(self.h), shape=(n,)).astype(dt).copy()

    @property
    def scale_factors(self):
        return self._tab(self.L.orc_scale_factors, self.nlevels + 1, np.float32)

    @property
    def inv_scale_factors(self):
        return self._tab(self.L.orc_inv_scale_factors, self.nlevels + 1, np.float32)

    @property
    def level_sigma2(self):
        return self._tab(self.L.orc_level_sigma2, self.nlevels + 1, np.float32)

    @property
    def inv_level_sigma2(self):
        return self._tab(self.L.orc_inv_level_sigma2, self.nlevels + 1, np.float32)

    @property
    def features_per_level(self):
        return self._tab(self.L.orc_features_per_level, self.nlevels + 1, np.int32)

    @property
    def umax(self):
        return self._tab(self.L.orc_umax, 16, np.int32)

    def extract(self, gray, cap=None):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        # (a level returns up to max(quota + 3, 4 * nIni) keypoints, nIni = round(W / H) roots: panoramic frames with tiny
        #  quotas exceed nfeatures by far more than 3 per level)
        cap = cap or (self.nfeatures + 64 * self.nlevels + 64)
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        rc = self.L.orc_extract(self.h, _p(gray), w, h, w, _p(kps), _p(desc), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError("orc_extract rc=%d" % rc)
        return kps[: n.value].copy(), desc[: n.value].copy()

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        self.L.orc_level_size(self.h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def _img(self, ptr, level):
        w, h = self.level_size(level)
        if not ptr:
            return None
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h, w)).copy()

    def level_image(self, level):
        return self._img(self.L.orc_level_image(self.h, level), level)

    def level_blurred(self, level):
        return self._img(self.L.orc_level_blurred(self.h, level), level)

    def level_candidates(self, level):
        p = C.c_void_p()
        n = self.L.orc_level_candidates(self.h, level, C.byref(p))
        if n <= 0:
            return np.zeros(0, CAND_DTYPE)
        buf = (C.c_char * (n * CAND_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=CAND_DTYPE, count=n).copy()

    def level_keypoints(self, level):
        return self.L.orc_level_keypoints(self.h, level)


def simd_available():
    """True when this CPU runs the SIMD variants (oracle/orb_simd.c, AVX2)."""
    return bool(lib().orc_simd_available())


def resize_linear(src, dw, dh, simd=False):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8_ex(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw, 1 if simd else 0)
    return dst


def fast9_nms(img, threshold, simd=False):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(cap, CAND_DTYPE)
    fn = lib().orc_fast9_nms_simd if simd else lib().orc_fast9_nms
    n = fn(_p(img), img.shape[1], img.shape[0], img.shape[1], threshold, _p(out), cap)
    return out[:n].copy()


def fast9_score_map(img, threshold, simd=False):
    """FAST-9/16 corner score of every pixel before NMS (0 = no corner at `threshold`)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    (lib().orc_fast9_score_map_simd if simd else lib().orc_fast9_score_map)(_p(img), w, h, w, int(threshold), _p(out))
    return out


def distribute_octtree(cand, minX, maxX, minY, maxY, N):
    cand = np.ascontiguousarray(cand, CAND_DTYPE)
    out = np.zeros(len(cand) + 4, np.int32)
    n = lib().orc_distribute_octtree(_p(cand), len(cand), minX, maxX, minY, maxY, N, _p(out), len(out))
    if n < 0:
        raise RuntimeError("orc_distribute_octtree rc=%d" % n)
    return out[:n].copy()


def gaussian_blur7(img, tie_mode=0, simd=False):
    img = np.ascontiguousarray(img, np.uint8)
    dst = np.zeros_like(img)
    lib().orc_gaussian_blur7_ex(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(dst), img.shape[1], tie_mode, 1 if simd else 0)
    return dst


def fast_atan2(y, x):
    return lib().orc_fast_atan2(float(y), float(x))


def ic_angle(img, x, y, umax):
    img = np.ascontiguousarray(img, np.uint8)
    um = np.ascontiguousarray(umax, np.int32)
    return lib().orc_ic_angle(C.c_void_p(img.ctypes.data), img.shape[1], x, y,
                              um.ctypes.data_as(C.POINTER(C.c_int32)))


def sincos_f(angle):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos_f(float(angle), C.byref(s), C.byref(c))
    return s.value, c.value


def orb_descriptor(blurred, x, y, angle_deg):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orc_orb_descriptor(_p(blurred), blurred.shape[1], x, y, float(angle_deg), _p(d))
    return d


def ingest_geometry(img, rotate_degrees=0, vertical_flip=False, horizontal_flip=False):
    """The reader's rotation + flips (image_sequence_reader.cc:186-205, :53-58) on [H, W] or [H, W, C]."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    oh, ow = (w, h) if rotate_degrees in (90, 270) else (h, w)
    out = np.zeros((oh, ow) if img.ndim == 2 else (oh, ow, cn), np.uint8)
    if lib().orc_ingest_geometry(_p(img), w, h, cn, int(rotate_degrees), int(bool(vertical_flip)), int(bool(horizontal_flip)), _p(out)) != 0:
        raise ValueError("unsupported rotation angle")
    return out


def rgb_to_gray(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    g = np.zeros((h, w), np.uint8)
    lib().orc_rgb_to_gray(_p(rgb), w, h, 3 * w, _p(g), w)
    return g


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def hamming_matrix(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    out = np.zeros((len(a), len(b)), np.uint16)
    lib().orc_hamming_matrix(_p(a), len(a), _p(b), len(b), _p(out))
    return out


def hamming_best2(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    bi = np.zeros(len(a), np.int32)
    b1 = np.zeros(len(a), np.uint16)
    b2 = np.zeros(len(a), np.uint16)
    lib().orc_hamming_best2(_p(a), len(a), _p(b), len(b), _p(bi), _p(b1), _p(b2))
    return bi, b1, b2


# ---------------------------------------------------------------- DBoW2 vocabulary oracle
class VocabOracle:
    """TemplatedVocabulary<FORB> restated (oracle/bow_oracle.c)."""

    def __init__(self, text_path):
        self.L_ = lib()
        self.h = self.L_.orc_vocab_load_text(text_path.encode())
        if not self.h:
            raise ValueError("cannot load vocabulary %s" % text_path)
        k, L, nn, nw = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self.L_.orc_vocab_info(self.h, C.byref(k), C.byref(L), C.byref(nn), C.byref(nw))
        self.k, self.L, self.nnodes, self.nwords = k.value, L.value, nn.value, nw.value

    def __del__(self):
        if getattr(self, "h", None):
            self.L_.orc_vocab_free(self.h)
            self.h = None

    def transform_features(self, desc, levelsup):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        word = np.zeros(n, np.uint32)
        weight = np.zeros(n, np.float64)
        node = np.zeros(n, np.uint32)
        for i in range(n):
            self.L_.orc_bow_transform_one(self.h, C.c_void_p(desc[i].ctypes.data), levelsup,
                                          C.c_void_p(word[i:].ctypes.data), C.c_void_p(weight[i:].ctypes.data),
                                          C.c_void_p(node[i:].ctypes.data))
        return word, weight, node

    def transform(self, desc, levelsup):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        bid, bval = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.float64)
        fnode, fstart, ffeat = np.zeros(n + 1, np.uint32), np.zeros(n + 2, np.int32), np.zeros(n + 1, np.uint32)
        nb, nf = C.c_int32(), C.c_int32()
        self.L_.orc_bow_transform(self.h, _p(desc), n, levelsup, _p(bid), _p(bval), C.cast(C.byref(nb), C.c_void_p),
                                  _p(fnode), _p(fstart), _p(ffeat), C.cast(C.byref(nf), C.c_void_p))
        nb, nf = nb.value, nf.value
        return (bid[:nb].copy(), bval[:nb].copy()), (fnode[:nf].copy(), fstart[:nf + 1].copy(), ffeat[:fstart[nf]].copy())


def bow_score_l1(a, b):
    (i1, v1), (i2, v2) = a, b
    i1, i2 = np.ascontiguousarray(i1, np.uint32), np.ascontiguousarray(i2, np.uint32)
    v1, v2 = np.ascontiguousarray(v1, np.float64), np.ascontiguousarray(v2, np.float64)
    return lib().orc_bow_score_l1(_p(i1), _p(v1), len(i1), _p(i2), _p(v2), len(i2))


_REF = os.path.join(_HERE, "_ref", "libdbow2_ref.so")


def ref_dbow2():
    """The REAL reference BowVector/FeatureVector code (oracle/Makefile.ref), or None."""
    if not os.path.exists(_REF):
        if os.path.isdir("/root/reference"):
            subprocess.check_call(["make", "-C", _HERE, "-f", "Makefile.ref"], stdout=subprocess.DEVNULL)
        else:
            return None
    R = C.CDLL(_REF)
    R.ref_bow_vectors.argtypes = [C.c_int] + [C.c_void_p] * 10
    R.ref_bow_vectors2.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 7
    return R


def ref_bow_vectors(word, weight, node, scoring=None, weighting=None):
    R = ref_dbow2()
    n = len(word)
    word = np.ascontiguousarray(word, np.uint32)
    weight = np.ascontiguousarray(weight, np.float64)
    node = np.ascontiguousarray(node, np.uint32)
    bid, bval = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.float64)
    fnode, fstart, ffeat = np.zeros(n + 1, np.uint32), np.zeros(n + 2, np.int32), np.zeros(n + 1, np.uint32)
    nb, nf = C.c_int32(), C.c_int32()
    if scoring is None:
        R.ref_bow_vectors(n, _p(word), _p(weight), _p(node), _p(bid), _p(bval), C.cast(C.byref(nb), C.c_void_p),
                          _p(fnode), _p(fstart), _p(ffeat), C.cast(C.byref(nf), C.c_void_p))
    else:
        rc = R.ref_bow_vectors2(n, _p(word), _p(weight), _p(node), int(scoring), int(weighting), _p(bid), _p(bval),
                                C.cast(C.byref(nb), C.c_void_p), _p(fnode), _p(fstart), _p(ffeat),
                                C.cast(C.byref(nf), C.c_void_p))
        assert rc == 0
    nb, nf = nb.value, nf.value
    return (bid[:nb].copy(), bval[:nb].copy()), (fnode[:nf].copy(), fstart[:nf + 1].copy(), ffeat[:fstart[nf]].copy())


# ---------------------------------------------------------------- Frame grid + guided matcher oracle
def frame_grid(kps, bounds):
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    start = np.zeros(64 * 48 + 1, np.int32)
    idx = np.zeros(max(len(kps), 1), np.int32)
    lib().orc_frame_grid(_p(kps), len(kps), bounds[0], bounds[1], bounds[2], bounds[3], _p(start), _p(idx))
    return start, idx[:start[-1]].copy()


def features_in_area(kps, grid, bounds, x, y, r, min_level, max_level):
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    out = np.zeros(len(kps) + 1, np.int32)
    n = lib().orc_features_in_area(_p(kps), _p(grid[0]), _p(grid[1]), bounds[0], bounds[1], bounds[2], bounds[3],
                                   x, y, r, min_level, max_level, _p(out))
    return out[:n].copy()


def search_for_initialization(kps1, desc1, kps2, desc2, bounds, prev_matched, window_size=100, nnratio=0.9,
                              check_orientation=True):
    kps1 = np.ascontiguousarray(kps1, KEYPOINT_DTYPE)
    kps2 = np.ascontiguousarray(kps2, KEYPOINT_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8)
    desc2 = np.ascontiguousarray(desc2, np.uint8)
    g2 = frame_grid(kps2, bounds)
    idx = np.ascontiguousarray(np.concatenate([g2[1], np.zeros(1, np.int32)]))
    prev = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.full(max(len(kps1), 1), -1, np.int32)
    nm = lib().orc_search_for_initialization(_p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2), len(kps2),
                                             _p(g2[0]), _p(idx), bounds[0], bounds[1], bounds[2], bounds[3],
                                             _p(prev), _p(m12), window_size, nnratio, int(check_orientation))
    return nm, m12[:len(kps1)].copy(), prev


def _prep(kps, desc, bounds, kp_has_point):
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    g = frame_grid(kps, bounds)
    idx = np.ascontiguousarray(np.concatenate([g[1], np.zeros(1, np.int32)]))
    has = np.ascontiguousarray(kp_has_point if kp_has_point is not None else np.zeros(len(kps), np.uint8), np.uint8)
    return kps, desc, g[0], idx, has


def search_by_projection_points(kps, desc, bounds, scale_factors, kp_has_point, valid, proj_x, proj_y, level, view_cos,
                                pdesc, pobs, th, nnratio):
    kps, desc, gs, gi, has = _prep(kps, desc, bounds, kp_has_point)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    a = [np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(proj_x, np.float32), np.ascontiguousarray(proj_y, np.float32),
         np.ascontiguousarray(level, np.int32), np.ascontiguousarray(view_cos, np.float32), np.ascontiguousarray(pdesc, np.uint8),
         np.ascontiguousarray(pobs, np.uint8)]
    out = np.full(max(len(kps), 1), -1, np.int32)
    nm = lib().orc_search_by_projection_points(_p(kps), _p(desc), len(kps), _p(gs), _p(gi), *bounds, _p(sf), _p(has), len(a[0]),
                                               *[_p(x) for x in a], th, nnratio, _p(out))
    return nm, out[:len(kps)].copy()


def search_by_projection_frame(kps, desc, bounds, scale_factors, kp_has_point, valid, u, v, last_octave, last_angle, pdesc,
                               pobs, th, check_orientation=True):
    kps, desc, gs, gi, has = _prep(kps, desc, bounds, kp_has_point)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    a = [np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32),
         np.ascontiguousarray(last_octave, np.int32), np.ascontiguousarray(last_angle, np.float32),
         np.ascontiguousarray(pdesc, np.uint8), np.ascontiguousarray(pobs, np.uint8)]
    out = np.full(max(len(kps), 1), -1, np.int32)
    nm = lib().orc_search_by_projection_frame(_p(kps), _p(desc), len(kps), _p(gs), _p(gi), *bounds, _p(sf), _p(has), len(a[0]),
                                              *[_p(x) for x in a], th, int(check_orientation), _p(out))
    return nm, out[:len(kps)].copy()


def log_f(x):
    """PredictScale's logf under the parity contract (match_oracle.c: orc_log_f)."""
    return float(lib().orc_log_f(float(x)))


def predict_scale(max_distance, current_dist, log_scale_factor, nlevels):
    """MapPoint::PredictScale (src/MapPoint.cc:516-531)."""
    return int(lib().orc_predict_scale(float(max_distance), float(current_dist), float(log_scale_factor), int(nlevels)))


def search_by_projection_keyframe(kps, desc, bounds, scale_factors, kp_has_point, valid, found, u, v, dist3d, min_distance,
                                  max_distance, log_scale_factor, kf_angle, pdesc, th, orb_dist, check_orientation=True):
    """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1476-1603), the
    relocalisation search; scale_factors has nlevels + 1 entries like the extractor's table."""
    kps, desc, gs, gi, has = _prep(kps, desc, bounds, kp_has_point)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    a = [np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(found, np.uint8), np.ascontiguousarray(u, np.float32),
         np.ascontiguousarray(v, np.float32), np.ascontiguousarray(dist3d, np.float32), np.ascontiguousarray(min_distance, np.float32),
         np.ascontiguousarray(max_distance, np.float32), np.ascontiguousarray(kf_angle, np.float32), np.ascontiguousarray(pdesc, np.uint8)]
    out = np.full(max(len(kps), 1), -1, np.int32)
    nm = lib().orc_search_by_projection_keyframe(_p(kps), _p(desc), len(kps), _p(gs), _p(gi), *bounds, _p(sf), len(sf) - 1,
                                                 float(log_scale_factor), _p(has), len(a[0]), *[_p(x) for x in a],
                                                 float(th), int(orb_dist), int(check_orientation), _p(out))
    return nm, out[:len(kps)].copy()


def search_by_bow(kf_desc, kf_angle, kf_valid, kf_fv, f_desc, f_angle, f_fv, nnratio, check_orientation=True):
    kf_desc = np.ascontiguousarray(kf_desc, np.uint8); f_desc = np.ascontiguousarray(f_desc, np.uint8)
    kf_angle = np.ascontiguousarray(kf_angle, np.float32); f_angle = np.ascontiguousarray(f_angle, np.float32)
    kf_valid = np.ascontiguousarray(kf_valid, np.uint8)
    A = [np.ascontiguousarray(kf_fv[0], np.uint32), np.ascontiguousarray(kf_fv[1], np.int32), np.ascontiguousarray(kf_fv[2], np.uint32)]
    B = [np.ascontiguousarray(f_fv[0], np.uint32), np.ascontiguousarray(f_fv[1], np.int32), np.ascontiguousarray(f_fv[2], np.uint32)]
    out = np.full(max(len(f_desc), 1), -1, np.int32)
    nm = lib().orc_search_by_bow(_p(kf_desc), _p(kf_angle), _p(kf_valid), len(kf_desc), _p(A[0]), _p(A[1]), _p(A[2]), len(A[0]),
                                 _p(f_desc), _p(f_angle), len(f_desc), _p(B[0]), _p(B[1]), _p(B[2]), len(B[0]),
                                 nnratio, int(check_orientation), _p(out))
    return nm, out[:len(f_desc)].copy()


def undistort_keypoints(kps, camera, dist):
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    cam = np.ascontiguousarray(camera, np.float32); d = np.ascontiguousarray(dist, np.float32)
    out = np.zeros_like(kps)
    lib().orc_undistort_keypoints(_p(kps), len(kps), _p(cam), _p(d), _p(out))
    return out


def image_bounds(cols, rows, camera, dist):
    cam = np.ascontiguousarray(camera, np.float32); d = np.ascontiguousarray(dist, np.float32)
    b = np.zeros(4, np.float32)
    lib().orc_image_bounds(cols, rows, _p(cam), _p(d), _p(b))
    return tuple(float(x) for x in b)


# ---- trajectory post-processing (post_oracle.c) ----

def _d(a, shape=None):
    a = np.ascontiguousarray(a, np.float64)
    return a if shape is None else a.reshape(shape)


def smooth_heading_directions(quat_wxyz, sigma):
    q = _d(quat_wxyz).reshape(-1, 4).copy()
    rc = lib().porc_smooth_heading_directions(_p(q), len(q), int(sigma))
    if rc:
        raise ValueError("CHECK_GT(sigma, 0)")
    return q


def smooth_time_series(values, times, targets, sigma):
    v, t, g = _d(values), _d(times), _d(targets)
    assert len(v) == len(t)
    out = np.zeros(len(g), np.float64)
    f = lib().porc_smooth_time_series
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    if f(_p(v), _p(t), len(v), _p(g), len(g), float(sigma), _p(out)):
        raise ValueError("CHECK_GT(sigma, 0)")
    return out


def trajectory_pca(translations):
    t = _d(translations).reshape(-1, 3)
    vec, val, mean = np.zeros((3, 3)), np.zeros(3), np.zeros(3)
    if lib().porc_trajectory_pca(_p(t), len(t), _p(vec), _p(val), _p(mean)):
        raise ValueError("fewer than 3 samples")
    return vec, val, mean


def project_directions(quat_wxyz, plane):
    q, pl = _d(quat_wxyz).reshape(-1, 4), _d(plane).reshape(2, 3)
    out = np.zeros((len(q), 2))
    lib().porc_project_directions(_p(q), len(q), _p(pl), _p(out))
    return out


def project_translations(translations, plane):
    t, pl = _d(translations).reshape(-1, 3).copy(), _d(plane).reshape(2, 3)
    lib().porc_project_translations(_p(t), len(t), _p(pl))
    return t


def turn_angles(dirs):
    d = _d(dirs).reshape(-1, 2)
    out = np.zeros(len(d))
    lib().porc_turn_angles(_p(d), len(d), _p(out))
    return out


# ---- fit_motion velocity calibration (calib_oracle.c; SURVEY §8 f4, BASELINE configs[4]) ----

def _series(gps_v, gps_t, rot, rot_t, acc, acc_t):
    a = [_d(gps_v), np.ascontiguousarray(gps_t, np.int64), _d(rot).reshape(-1, 3), np.ascontiguousarray(rot_t, np.int64),
         _d(acc).reshape(-1, 3), np.ascontiguousarray(acc_t, np.int64)]
    assert len(a[0]) == len(a[1]) and len(a[2]) == len(a[3]) and len(a[4]) == len(a[5])
    return a, [_p(a[0]), _p(a[1]), len(a[0]), _p(a[2]), _p(a[3]), len(a[2]), _p(a[4]), _p(a[5]), len(a[4])]


_SER = [C.c_void_p, C.c_void_p, C.c_int] * 3


def calibrator_eval(gps_v, gps_t, rot, rot_t, acc, acc_t, x):
    keep, args = _series(gps_v, gps_t, rot, rot_t, acc, acc_t)
    x = _d(x); fx = C.c_double(0); g = np.zeros(9)
    f = lib().porc_calibrator_eval
    f.argtypes = _SER + [C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
    if f(*args, _p(x), C.byref(fx), _p(g)):
        raise ValueError("calibrator cannot be built (CHECK failure in the reference)")
    return fx.value, g


def fit_windows(gps_v, gps_t, rot, rot_t, acc, acc_t, batch_size=40, shift_step=5, max_iters=500):
    keep, args = _series(gps_v, gps_t, rot, rot_t, acc, acc_t)
    nw = lib().porc_num_windows(len(keep[0]), shift_step)
    x, res, it = np.zeros((nw, 9)), np.zeros(nw), np.zeros(nw, np.int32)
    f = lib().porc_fit_windows
    f.argtypes = _SER + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    if f(*args, batch_size, shift_step, max_iters, _p(x), _p(res), _p(it)) != nw:
        raise ValueError("calibrator cannot be built (CHECK failure in the reference)")
    return x, res, it


def fit_motion_velocities(gps_v, gps_t, rot, rot_t, acc, acc_t, vertical_axis, batch_size=40, shift_step=5, max_iters=500,
                          post_smoothing_sigma_sec=0.003, min_velocity=5.0, min_rotation_rad=0.2):
    keep, args = _series(gps_v, gps_t, rot, rot_t, acc, acc_t)
    cap = len(keep[2]) + len(keep[4])
    t, v, fwd, va = np.zeros(cap, np.int64), np.zeros(cap), np.zeros(3), _d(vertical_axis)
    f = lib().porc_fit_motion_velocities
    f.argtypes = _SER + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    n = f(*args, _p(va), batch_size, shift_step, max_iters, post_smoothing_sigma_sec, min_velocity, min_rotation_rad, _p(t), _p(v), _p(fwd))
    if n < 0:
        raise ValueError("fit_motion oracle failed: %d" % n)
    return t[:n].copy(), v[:n].copy(), fwd


def principal_rotation_axes(rot, rot_t, integration_interval_usec=500000):
    r, t = _d(rot).reshape(-1, 3), np.ascontiguousarray(rot_t, np.int64)
    vec = np.zeros((3, 3))
    f = lib().porc_principal_rotation_axes
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]
    rc = f(_p(r), _p(t), len(r), int(integration_interval_usec), _p(vec))
    if rc:
        raise ValueError("CHECK failure in GetPrincipalRotationAxes (%d)" % rc)
    return vec


def angular_velocities_around_axis(rot, axis):
    r, a = _d(rot).reshape(-1, 3), _d(axis)
    out = np.zeros(len(r))
    if lib().porc_angular_velocities_around_axis(_p(r), len(r), _p(a), _p(out)):
        raise ValueError("axis not normalised")
    return out


def kahan_sum(values):
    v = _d(values); v = v.reshape(len(v), -1)
    out = np.zeros(v.shape[1])
    lib().porc_kahan_sum(_p(v), v.shape[0], v.shape[1], _p(out))
    return out


_MATH_REF = os.path.join(_HERE, "_ref", "libmath_ref.so")


def ref_kahan_sum(values):
    """The REFERENCE's own KahanSum (oracle/_ref/libmath_ref.so, built from include/math/math.hpp by Makefile.ref)."""
    if not os.path.exists(_MATH_REF):
        if os.path.isdir("/root/reference"):
            subprocess.check_call(["make", "-C", _HERE, "-f", "Makefile.ref"], stdout=subprocess.DEVNULL)
        else:
            return None
    v = _d(values); v = v.reshape(len(v), -1)
    out = np.zeros(v.shape[1])
    C.CDLL(_MATH_REF).ref_kahan_sum(_p(v), v.shape[0], v.shape[1], _p(out))
    return out

"""Host-side mirror of the reference's ORB interface over the C ABI.

Names, argument meaning and error behaviour follow
  ORB_SLAM2::ORBextractor   thirdparty/orb-slam2/include/ORBextractor.h:44-110
  ORB_SLAM2::ORBmatcher     thirdparty/orb-slam2/include/ORBmatcher.h:44 (DescriptorDistance)
so the parity tests read like tests of the reference classes.  All compute happens in
libpgorb.so (HIP, gfx950); numpy / torch only carry buffers.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28


def _p(a):
    return C.c_void_p(a.ctypes.data)


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) (ORBextractor.h:51-52).

    Extra keyword arguments size the device context (largest frame, frames per batch, GPU).
    """

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, *,
                 max_width=1920, max_height=1080, max_batch=1, device=0, blur_tie_mode=0):
        self._L = _lib.lib()
        self.nfeatures, self.nlevels = int(nfeatures), int(nlevels)
        self.scaleFactor = float(scaleFactor)
        self.max_batch = int(max_batch)
        self.device = int(device)
        prm = _lib.PgorbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST,
                               max_width, max_height, max_batch, device, blur_tie_mode)
        h = C.c_void_p()
        rc = self._L.pgorb_create(C.byref(prm), C.byref(h))
        if rc != 0:
            raise _lib.PgorbError(rc, self._L.pgorb_last_error(None).decode())
        self._h = h
        self._streams = weakref.WeakSet()             # FrameStreams of this context: pgorb_destroy takes them along

    def close(self):
        if getattr(self, "_h", None):
            for st in list(self._streams):            # the C context destroys its live streams: their handles die with it
                st._s = None
            self._L.pgorb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise _lib.PgorbError(rc, self._L.pgorb_last_error(self._h).decode())
        return rc

    # ---- getters, ORBextractor.h:63-83 -------------------------------------------------
    def GetLevels(self):
        return self._L.pgorb_levels(self._h)

    def GetScaleFactor(self):
        return self.scaleFactor

    def _tables(self):
        n = self.nlevels + 1
        t = [np.zeros(n, np.float32) for _ in range(4)]
        fp = C.POINTER(C.c_float)
        self._check(self._L.pgorb_scale_tables(self._h, *[a.ctypes.data_as(fp) for a in t]))
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        out = np.zeros(self.nlevels + 1, np.int32)
        self._check(self._L.pgorb_features_per_level(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def max_keypoints(self, w, h):
        return self._check(self._L.pgorb_max_keypoints(self._h, w, h))

    # ---- operator(), ORBextractor.h:59-61 -----------------------------------------------
    def __call__(self, image, mask=None):
        """(keypoints, descriptors) of one CV_8UC1 image.  `mask` is ignored, as in the
        reference (ORBextractor.h:58).  An empty image returns empty outputs (:1045)."""
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise TypeError("image must be CV_8UC1 (2-D uint8)")      # assert(type==CV_8UC1) :1049
        out = self.extract_batch([image])
        return out[0]

    def extract_batch(self, frames):
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        h, w = frames[0].shape
        cap = self.max_keypoints(w, h)
        nfr = len(frames)
        kps = np.zeros((nfr, cap), KEYPOINT_DTYPE)
        desc = np.zeros((nfr, cap, 32), np.uint8)
        n = np.zeros(nfr, np.int32)
        ptrs = (C.c_void_p * nfr)(*[f.ctypes.data for f in frames])
        self._check(self._L.pgorb_extract_batch(self._h, ptrs, nfr, w, h, w, _p(kps), _p(desc), cap,
                                                n.ctypes.data_as(C.POINTER(C.c_int32))))
        return [(kps[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(nfr)]

    def extract_batch_device(self, frames_u8, kps_out=None, desc_out=None, n_out=None, stream=None):
        """Resident path: `frames_u8` is a CUDA(HIP) torch uint8 tensor [B, H, W].  Returns torch
        tensors (kps [B,cap,7] float32 view of pgorb_keypoint, desc [B,cap,32] u8, n [B] i32);
        asynchronous on `stream` (default: torch's current stream)."""
        import torch
        B, H, W = frames_u8.shape
        cap = self.max_keypoints(W, H)
        dev = frames_u8.device
        if kps_out is None:
            kps_out = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)
            desc_out = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
            n_out = torch.empty((B,), dtype=torch.int32, device=dev)
        s = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        self._check(self._L.pgorb_extract_batch_device(
            self._h, C.c_void_p(frames_u8.data_ptr()), B, W, H, frames_u8.stride(1),
            frames_u8.stride(0), C.c_void_p(kps_out.data_ptr()), C.c_void_p(desc_out.data_ptr()),
            cap, C.c_void_p(n_out.data_ptr()), C.c_void_p(s)))
        return kps_out, desc_out, n_out

    def extract_batch_color_device(self, frames_rgb, rgb_order=True, stream=None):
        """`frames_rgb`: CUDA(HIP) torch uint8 tensor [B, H, W, C], C = 3 or 4; fuses
        Tracking::GrabImageMonocular's cvtColor (Tracking.cc:247-260) in front of the extractor."""
        import torch
        B, H, W, Cn = frames_rgb.shape
        cap = self.max_keypoints(W, H)
        dev = frames_rgb.device
        kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)
        desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
        n = torch.empty((B,), dtype=torch.int32, device=dev)
        s = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        self._check(self._L.pgorb_extract_batch_color_device(
            self._h, C.c_void_p(frames_rgb.data_ptr()), B, W, H, frames_rgb.stride(1), frames_rgb.stride(0), Cn,
            int(bool(rgb_order)), C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()), cap,
            C.c_void_p(n.data_ptr()), C.c_void_p(s)))
        return kps, desc, n

    def extract_batch_ingest_device(self, frames, rgb_order=True, rotate_degrees=0, vertical_flip=False,
                                    horizontal_flip=False, stream=None):
        """`frames`: CUDA(HIP) torch uint8 tensor [B, H, W] (grey) or [B, H, W, C] (C = 3 or 4) exactly
        as decoded; applies the reader's rotation (0/90/180/270, image_sequence_reader.cc:186-205)
        and flips (:53-58) and Tracking's grey conversion on the device, then extracts."""
        import torch
        if frames.dim() == 3:
            B, H, W = frames.shape
            Cn = 1
        else:
            B, H, W, Cn = frames.shape
        ow, oh = (H, W) if rotate_degrees in (90, 270) else (W, H)
        cap = self.max_keypoints(ow, oh)
        dev = frames.device
        kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)
        desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
        n = torch.empty((B,), dtype=torch.int32, device=dev)
        s = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        self._check(self._L.pgorb_extract_batch_ingest_device(
            self._h, C.c_void_p(frames.data_ptr()), B, W, H, frames.stride(1), frames.stride(0), Cn,
            int(bool(rgb_order)), int(rotate_degrees), int(bool(vertical_flip)), int(bool(horizontal_flip)),
            C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()), cap, C.c_void_p(n.data_ptr()), C.c_void_p(s)))
        return kps, desc, n

    def check_async(self, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        self._check(self._L.pgorb_check_async(self._h, C.c_void_p(s)))

    def match_batch_device(self, desc, n, pair_query, pair_train, out=None, stream=None):
        """best/second-best Hamming match of frame pair_query[p] against pair_train[p]."""
        import torch
        B, cap, _ = desc.shape
        npairs = pair_query.numel()
        dev = desc.device
        if out is None:
            out = (torch.empty((npairs, cap), dtype=torch.int32, device=dev),
                   torch.empty((npairs, cap), dtype=torch.int16, device=dev),
                   torch.empty((npairs, cap), dtype=torch.int16, device=dev))
        s = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        self._check(self._L.pgorb_match_batch_device(
            self._h, C.c_void_p(desc.data_ptr()), C.c_void_p(n.data_ptr()), cap,
            C.c_void_p(pair_query.data_ptr()), C.c_void_p(pair_train.data_ptr()), npairs,
            C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
            C.c_void_p(out[2].data_ptr()), C.c_void_p(s)))
        return out

    def log_scale_factor(self):
        """Frame::mfLogScaleFactor = log(mfScaleFactor) (Frame.cc:188) under the library's log contract."""
        return float(self._L.pgorb_log_scale_factor(self._h))

    def predict_scale(self, max_distance, current_dist):
        """MapPoint::PredictScale(currentDist, pF) (src/MapPoint.cc:516-531)."""
        return self._check(self._L.pgorb_predict_scale(self._h, float(max_distance), float(current_dist)))

    def set_option(self, key, value):
        """Measurement switches of the library (include/pgorb.h: pgorb_set_option)."""
        self._check(self._L.pgorb_set_option(self._h, key.encode(), int(value)))

    def matcher_name(self, cap_per_frame):
        return "popcount" if self._L.pgorb_matcher_is_popcount(self._h, int(cap_per_frame)) else "mfma_fp4"

    def get_option(self, key):
        v = self._L.pgorb_get_option(self._h, key.encode())
        if v == -2147483648:                                     # PGORB_OPTION_UNKNOWN
            raise KeyError(key)
        return v

    def fast_kernel_name(self):
        """Name of the K2 kernel (for bench.py's roofline object)."""
        return "k_fast_cells"

    STAGES = ("pyramid", "fast", "quadtree", "describe", "match")

    def profile_begin(self, max_calls):
        self._check(self._L.pgorb_profile_begin(self._h, max_calls))

    def profile_read(self):
        """{stage: mean ms per call} measured with HIP events on the launch stream."""
        ms = (C.c_double * 5)()
        n = self._check(self._L.pgorb_profile_read(self._h, ms))
        return n, dict(zip(self.STAGES, list(ms)))

    # ---- stage taps (parity tests) --------------------------------------------------------
    def debug_level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        self._check(self._L.pgorb_debug_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def debug_level_image(self, frame, level):
        w, h = self.debug_level_size(level)
        out = np.zeros((h, w), np.uint8)
        self._check(self._L.pgorb_debug_level_image(self._h, frame, level, _p(out)))
        return out

    def debug_level_candidates(self, frame, level):
        w, h = self.debug_level_size(level)
        cap = w * h // 2 + 64
        x, y, r = (np.zeros(cap, np.int32) for _ in range(3))
        n = self._check(self._L.pgorb_debug_level_candidates(self._h, frame, level, _p(x), _p(y), _p(r), cap))
        return x[:n].copy(), y[:n].copy(), r[:n].copy()

    def debug_level_keypoints(self, frame, level):
        return self._check(self._L.pgorb_debug_level_keypoints(self._h, frame, level))

    # ---- Hamming (device) -------------------------------------------------------------------
    def hamming_matrix(self, a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        out = np.zeros((len(a), len(b)), np.uint16)
        self._check(self._L.pgorb_hamming_matrix(self._h, _p(a), len(a), _p(b), len(b), _p(out)))
        return out

    def hamming_best2(self, a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        bi = np.zeros(len(a), np.int32)
        b1 = np.zeros(len(a), np.uint16)
        b2 = np.zeros(len(a), np.uint16)
        self._check(self._L.pgorb_hamming_best2(self._h, _p(a), len(a), _p(b), len(b), _p(bi), _p(b1), _p(b2)))
        return bi, b1, b2


class Frame:
    """The part of ORB_SLAM2::Frame the front end needs (thirdparty/orb-slam2/include/Frame.h):
    mvKeys (== mvKeysUndistorted for k1 == 0), mDescriptors, the image bounds and the 64x48 grid
    (AssignFeaturesToGrid, src/Frame.cc:234-249), built on the GPU."""

    def __init__(self, extractor, image, camera=None, dist_coef=None):
        """camera = (fx, fy, cx, cy), dist_coef = (k1, k2, p1, p2[, k3]) like mK / mDistCoef; without
        them (or with k1 == 0) the keypoints are used as they are (Frame.cc:410-414)."""
        self.ext = extractor
        self.mvKeys, self.mDescriptors = extractor(image)
        self.N = len(self.mvKeys)
        h, w = image.shape
        L = extractor._L
        if camera is not None and dist_coef is not None and float(dist_coef[0]) != 0.0:
            cam = np.ascontiguousarray(camera, np.float32)
            dc = np.zeros(5, np.float32)
            dc[:len(dist_coef)] = dist_coef
            self.mvKeysUndistorted = np.zeros_like(self.mvKeys)                      # UndistortKeyPoints (Frame.cc:408-438)
            if self.N:
                extractor._check(L.pgorb_undistort_keypoints(extractor._h, _p(self.mvKeys), self.N, _p(cam), _p(dc),
                                                             _p(self.mvKeysUndistorted)))
            b = np.zeros(4, np.float32)
            L.pgorb_image_bounds(w, h, _p(cam), _p(dc), _p(b))                       # ComputeImageBounds (Frame.cc:440-467)
            self.bounds = tuple(float(x) for x in b)
        else:
            self.mvKeysUndistorted = self.mvKeys
            self.bounds = (0.0, float(w), 0.0, float(h))      # mnMinX, mnMaxX, mnMinY, mnMaxY (Frame.cc:461-466)
        self.grid_start = np.zeros(64 * 48 + 1, np.int32)
        self.grid_idx = np.zeros(max(self.N, 1), np.int32)
        if self.N:
            extractor._check(L.pgorb_frame_grid(extractor._h, _p(self.mvKeysUndistorted), self.N, *self.bounds,
                                                _p(self.grid_start), _p(self.grid_idx)))

    def grid_cell(self, col, row):
        """mGrid[col][row] as an index array."""
        c = col * 48 + row
        return self.grid_idx[self.grid_start[c]:self.grid_start[c + 1]]


class MapPoints:
    """What the matchers read from a set of ORB_SLAM2::MapPoint (thirdparty/orb-slam2/include/
    MapPoint.h) as arrays: valid (mbTrackInView && !isBad()), projection, predicted level, viewing
    cosine, representative descriptor, Observations() > 0."""

    def __init__(self, valid, proj_x, proj_y, level, view_cos, descriptors, has_obs):
        self.valid = np.ascontiguousarray(valid, np.uint8)
        self.proj_x = np.ascontiguousarray(proj_x, np.float32)
        self.proj_y = np.ascontiguousarray(proj_y, np.float32)
        self.level = np.ascontiguousarray(level, np.int32)
        self.view_cos = np.ascontiguousarray(view_cos, np.float32)
        self.descriptors = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        self.has_obs = np.ascontiguousarray(has_obs, np.uint8)


class ORBmatcher:
    """ORBmatcher(nnratio, checkOri) (thirdparty/orb-slam2/include/ORBmatcher.h:40-44)."""

    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30               # src/ORBmatcher.cc:38-40

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)

    def SearchByProjection(self, F, points, th, kp_has_point=None):
        """SearchByProjection(Frame &F, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:46-131).
        Returns (nmatches, assigned) with assigned[i] = index into `points` written to
        F.mvpMapPoints[i], or -1."""
        ext = F.ext
        has = np.ascontiguousarray(kp_has_point if kp_has_point is not None else np.zeros(max(F.N, 1), np.uint8), np.uint8)
        out = np.full(max(F.N, 1), -1, np.int32)
        nm = ext._check(ext._L.pgorb_search_by_projection_points(
            ext._h, _p(F.mvKeysUndistorted), _p(F.mDescriptors), F.N, *F.bounds, _p(has), len(points.valid),
            _p(points.valid), _p(points.proj_x), _p(points.proj_y), _p(points.level), _p(points.view_cos),
            _p(points.descriptors), _p(points.has_obs), float(th), self.mfNNratio, _p(out)))
        return nm, out[:F.N].copy()

    def SearchByProjectionLastFrame(self, CurrentFrame, valid, u, v, last_octave, last_angle, point_desc, point_has_obs,
                                    th, kp_has_point=None):
        """The matching loop of SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th,
        bMono=true) (src/ORBmatcher.cc:1355-1474) for given projections (u, v)."""
        ext = CurrentFrame.ext
        F = CurrentFrame
        has = np.ascontiguousarray(kp_has_point if kp_has_point is not None else np.zeros(max(F.N, 1), np.uint8), np.uint8)
        a = [np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32),
             np.ascontiguousarray(last_octave, np.int32), np.ascontiguousarray(last_angle, np.float32),
             np.ascontiguousarray(point_desc, np.uint8), np.ascontiguousarray(point_has_obs, np.uint8)]
        out = np.full(max(F.N, 1), -1, np.int32)
        nm = ext._check(ext._L.pgorb_search_by_projection_frame(
            ext._h, _p(F.mvKeysUndistorted), _p(F.mDescriptors), F.N, *F.bounds, _p(has), len(a[0]),
            *[_p(x) for x in a], float(th), int(self.mbCheckOrientation), _p(out)))
        return nm, out[:F.N].copy()

    def SearchByProjectionKeyFrame(self, CurrentFrame, valid, already_found, u, v, dist3d, min_distance, max_distance, kf_angle,
                                   point_desc, th, ORBdist, kp_has_point=None):
        """The matching loop of SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist)
        (src/ORBmatcher.cc:1476-1603; Tracking::Relocalization) for given projections (u, v) and depths: the bounds / depth
        tests, MapPoint::PredictScale, the window search and the rotation histogram run on the device."""
        ext = CurrentFrame.ext
        F = CurrentFrame
        has = np.ascontiguousarray(kp_has_point if kp_has_point is not None else np.zeros(max(F.N, 1), np.uint8), np.uint8)
        a = [np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(already_found, np.uint8), np.ascontiguousarray(u, np.float32),
             np.ascontiguousarray(v, np.float32), np.ascontiguousarray(dist3d, np.float32), np.ascontiguousarray(min_distance, np.float32),
             np.ascontiguousarray(max_distance, np.float32), np.ascontiguousarray(kf_angle, np.float32), np.ascontiguousarray(point_desc, np.uint8)]
        out = np.full(max(F.N, 1), -1, np.int32)
        nm = ext._check(ext._L.pgorb_search_by_projection_keyframe(
            ext._h, _p(F.mvKeysUndistorted), _p(F.mDescriptors), F.N, *F.bounds, _p(has), len(a[0]), *[_p(x) for x in a],
            ext.log_scale_factor(), float(th), int(ORBdist), int(self.mbCheckOrientation), _p(out)))
        return nm, out[:F.N].copy()

    def SearchByBoW(self, ext, kf_desc, kf_angle, kf_point_valid, kf_featvec, F, f_featvec):
        """SearchByBoW(KeyFrame* pKF, Frame &F, vpMapPointMatches) (src/ORBmatcher.cc:161-290).
        Feature vectors are the (nodes, starts, features) triples of ORBVocabulary.transform().
        Returns (nmatches, matches) with matches[j] = key-frame keypoint index or -1."""
        kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
        ka = np.ascontiguousarray(kf_angle, np.float32)
        kv = np.ascontiguousarray(kf_point_valid, np.uint8)
        A = [np.ascontiguousarray(kf_featvec[0], np.uint32), np.ascontiguousarray(kf_featvec[1], np.int32),
             np.ascontiguousarray(kf_featvec[2], np.uint32)]
        B = [np.ascontiguousarray(f_featvec[0], np.uint32), np.ascontiguousarray(f_featvec[1], np.int32),
             np.ascontiguousarray(f_featvec[2], np.uint32)]
        fa = np.ascontiguousarray(F.mvKeys["angle"], np.float32)
        out = np.full(max(F.N, 1), -1, np.int32)
        nm = ext._check(ext._L.pgorb_search_by_bow(
            ext._h, _p(kd), _p(ka), _p(kv), len(kd), _p(A[0]), _p(A[1]), _p(A[2]), len(A[0]),
            _p(F.mDescriptors), _p(fa), F.N, _p(B[0]), _p(B[1]), _p(B[2]), len(B[0]),
            self.mfNNratio, int(self.mbCheckOrientation), _p(out)))
        return nm, out[:F.N].copy()

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10):
        """(nmatches, vnMatches12); vbPrevMatched ([N1,2] float32) is updated in place, as in
        src/ORBmatcher.cc:407-522."""
        ext = F1.ext
        prev = np.ascontiguousarray(vbPrevMatched, np.float32)
        m12 = np.full(max(F1.N, 1), -1, np.int32)
        nm = ext._check(ext._L.pgorb_search_for_initialization(
            ext._h, _p(F1.mvKeysUndistorted), _p(F1.mDescriptors), F1.N,
            _p(F2.mvKeysUndistorted), _p(F2.mDescriptors), F2.N, *F2.bounds, _p(prev), _p(m12),
            int(windowSize), self.mfNNratio, int(self.mbCheckOrientation)))
        vbPrevMatched[...] = prev
        return nm, m12[:F1.N].copy()

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.frombuffer(bytes(a), np.uint8) if isinstance(a, (bytes, bytearray)) else a
        b = np.frombuffer(bytes(b), np.uint8) if isinstance(b, (bytes, bytearray)) else b
        a = np.ascontiguousarray(a, np.uint8).reshape(32)
        b = np.ascontiguousarray(b, np.uint8).reshape(32)
        return _lib.lib().pgorb_descriptor_distance(_p(a), _p(b))


class FrameStream:
    """Streamed ingest (include/pgorb.h, pgorb_stream_*): the frame loop around the extractor for frames that
    start in host memory -- ImageSequenceSource::next() -> System::TrackMonocular in the reference
    (src/slam/track_image_sequence.cc:43-47).  `depth` page-locked input slots of `batch` frames; submit() never
    blocks, wait() returns the batch's results as numpy views of page-locked memory."""

    def __init__(self, extractor, w, h, batch, depth=3, channels=1, rgb_order=True, rotate_degrees=0,
                 vertical_flip=False, horizontal_flip=False):
        """w x h: the frames AS DECODED (before rotation); channels 1 (grey), 3 (RGB24 / BGR24) or 4.  With anything but
        upright grey the slots hold the decoder's frames and rotation / flips / cvtColor run on the device in front of
        the pyramid (pgorb_stream_create_ingest; image_sequence_reader.cc:53-58,186-205, Tracking.cc:247-260)."""
        self.ext, self.w, self.h, self.batch, self.depth = extractor, int(w), int(h), int(batch), int(depth)
        self.channels = int(channels)
        self._L = extractor._L
        hs = C.c_void_p()
        extractor._check(self._L.pgorb_stream_create_ingest(extractor._h, self.w, self.h, self.channels, int(bool(rgb_order)),
                                                            int(rotate_degrees), int(bool(vertical_flip)), int(bool(horizontal_flip)),
                                                            self.batch, self.depth, C.byref(hs)))
        self._s = hs
        extractor._streams.add(self)

    def close(self):
        """Frees the page-locked slots: every array input() / wait() / frontend_results() returned is invalid afterwards."""
        if getattr(self, "_s", None):
            self._L.pgorb_stream_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, address, dtype, shape):
        """numpy view of page-locked memory the C stream owns.  The view keeps THIS object alive (its ctypes base holds a
        reference), so the memory is not freed by garbage collection while a result array is still in use; an explicit
        close() -- or submitting the slot again -- still invalidates it."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = (C.c_uint8 * nbytes).from_address(address)
        buf._pgorb_owner = self
        return np.frombuffer(buf, dtype).reshape(shape)

    def input(self, slot):
        """numpy view [batch, h, w] (grey) or [batch, h, w, channels] of the slot's page-locked input frames: the decoder
        writes its frames here."""
        p = self._L.pgorb_stream_input(self._s, slot)
        shape = (self.batch, self.h, self.w) if self.channels == 1 else (self.batch, self.h, self.w, self.channels)
        return self._view(p, np.uint8, shape)

    def reset(self):
        self.ext._check(self._L.pgorb_stream_reset(self._s))

    def frontend(self, bounds, window_size=100, nnratio=0.9, check_orientation=True, bow_levelsup=-1):
        """Enable the per-frame front-end stage: grid, SearchForInitialization(previous, current) and -- with
        bow_levelsup >= 0 and a vocabulary uploaded to the extractor -- the BoW transform, on the device per batch."""
        self.ext._check(self._L.pgorb_stream_frontend(self._s, *[float(b) for b in bounds], int(window_size), float(nnratio),
                                                      int(bool(check_orientation)), int(bow_levelsup)))
        self._fe_bow = bow_levelsup >= 0

    def frontend_results(self, slot, nframes, cap):
        """(matches12[frames, cap], nmatches[frames], word, weight, node [frames, cap] or None) of a collected slot."""
        ptr = [C.c_void_p() for _ in range(5)]
        self.ext._check(self._L.pgorb_stream_frontend_results(self._s, slot, *[C.byref(p) for p in ptr]))

        def view(p, dtype, shape):
            return self._view(p.value, dtype, shape) if p.value else None
        return (view(ptr[0], np.int32, (nframes, cap)), view(ptr[1], np.int32, (nframes,)), view(ptr[2], np.uint32, (nframes, cap)),
                view(ptr[3], np.float64, (nframes, cap)), view(ptr[4], np.uint32, (nframes, cap)))

    def submit(self, slot, nframes=None):
        self.ext._check(self._L.pgorb_stream_submit(self._s, slot, self.batch if nframes is None else int(nframes)))

    def wait(self, slot):
        """(n[frames], kps[frames, cap], desc[frames, cap, 32], best_idx, best, second [frames, cap]) views."""
        ptr = [C.c_void_p() for _ in range(6)]
        cap = C.c_int32()
        nf = self.ext._check(self._L.pgorb_stream_wait(self._s, slot, *[C.byref(p) for p in ptr], C.byref(cap)))
        cap = cap.value

        def view(p, dtype, shape):
            return self._view(p.value, dtype, shape)
        return (view(ptr[0], np.int32, (nf,)), view(ptr[1], KEYPOINT_DTYPE, (nf, cap)), view(ptr[2], np.uint8, (nf, cap, 32)),
                view(ptr[3], np.int32, (nf, cap)), view(ptr[4], np.uint16, (nf, cap)), view(ptr[5], np.uint16, (nf, cap)))


class DeviceFrameStream:
    """The device-resident form of the stream (include/pgorb.h, pgorb_stream_create_device): frames are torch CUDA(HIP)
    uint8 tensors [B, H, W], results stay on the device, and `lanes` batches are in flight INSIDE the library (slot k
    runs on lane k % lanes, each lane an independent extractor working set on its own HIP stream).  Results equal the
    one-batch-at-a-time calls, including the match of a batch's first frame against the previous batch's last."""

    def __init__(self, extractor, w, h, batch, depth=2, lanes=2):
        self.ext, self.w, self.h, self.batch, self.depth = extractor, int(w), int(h), int(batch), int(depth)
        self._L = extractor._L
        hs = C.c_void_p()
        extractor._check(self._L.pgorb_stream_create_device(extractor._h, self.w, self.h, self.batch, self.depth, int(lanes), C.byref(hs)))
        self._s = hs
        self._keep = [None] * self.depth              # the submitted frame tensors (level 0 may alias them)
        extractor._streams.add(self)

    def close(self):
        if getattr(self, "_s", None):
            self._L.pgorb_stream_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lanes(self):
        return self._L.pgorb_stream_lanes(self._s)

    def reset(self):
        self.ext._check(self._L.pgorb_stream_reset(self._s))

    def frontend(self, bounds, window_size=100, nnratio=0.9, check_orientation=True, bow_levelsup=-1):
        self.ext._check(self._L.pgorb_stream_frontend(self._s, *[float(b) for b in bounds], int(window_size), float(nnratio),
                                                      int(bool(check_orientation)), int(bow_levelsup)))

    def submit(self, slot, frames_u8, stream=None):
        """Queue one batch; returns at once.  `frames_u8` must be ready on `stream` (default: torch's current stream)."""
        import torch
        nb, h, w = frames_u8.shape
        s = stream if stream is not None else torch.cuda.current_stream(frames_u8.device).cuda_stream
        self.ext._check(self._L.pgorb_stream_submit_device(self._s, slot, C.c_void_p(frames_u8.data_ptr()), nb, frames_u8.stride(1),
                                                           frames_u8.stride(0), C.c_void_p(s)))
        self._keep[slot] = frames_u8

    def wait(self, slot, on_host=True, stream=None):
        """(n[frames], kps[frames, cap, 7] f32 view, desc[frames, cap, 32], best_idx, best, second [frames, cap]) as torch
        tensors that ALIAS the slot's device result block (valid until the slot is submitted again)."""
        import torch
        ptr = [C.c_void_p() for _ in range(6)]
        cap = C.c_int32()
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        nf = self.ext._check(self._L.pgorb_stream_wait_device(self._s, slot, int(bool(on_host)), C.c_void_p(s),
                                                              *[C.byref(p) for p in ptr], C.byref(cap)))
        cap = cap.value
        dev = torch.device("cuda", self.ext.device)

        def view(p, dtype, shape, itemsize):
            return _device_tensor(p.value, int(np.prod(shape)) * itemsize, dev).view(dtype).reshape(shape)
        return (view(ptr[0], torch.int32, (nf,), 4), view(ptr[1], torch.float32, (nf, cap, 7), 4), view(ptr[2], torch.uint8, (nf, cap, 32), 1),
                view(ptr[3], torch.int32, (nf, cap), 4), view(ptr[4], torch.int16, (nf, cap), 2), view(ptr[5], torch.int16, (nf, cap), 2))


def _device_tensor(address, nbytes, device):
    """A torch uint8 tensor over `nbytes` of device memory the library owns (no copy): through the CUDA array interface."""
    import torch

    class _Raw:
        pass
    r = _Raw()
    r.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(address), False), "version": 2}
    return torch.as_tensor(r, device=device)

/* include/pgorb.h -- C ABI of libpgorb.so, the MI355X (gfx950) ORB front end.
 *
 * Drop-in boundary for pilotguru's per-frame visual-odometry front end.  The
 * reference has no FFI layer; its seam is a C++ functor plus two helpers
 * (SURVEY.md section 8b).  Each entry point below names the reference
 * interface it replaces (paths relative to the reference tree):
 *
 *   pgorb_create / pgorb_destroy     ORBextractor::ORBextractor(nfeatures, scaleFactor,
 *                                    nlevels, iniThFAST, minThFAST)
 *                                    thirdparty/orb-slam2/include/ORBextractor.h:51-54,
 *                                    constructed at src/Tracking.cc:137-143
 *   pgorb_extract*                   ORBextractor::operator()(image, mask, keypoints,
 *                                    descriptors)  include/ORBextractor.h:59-61,
 *                                    called from Frame::ExtractORB src/Frame.cc:251-257
 *   pgorb_scale_tables, _levels      GetScaleFactors()/GetInverseScaleFactors()/
 *                                    GetScaleSigmaSquares()/GetInverseScaleSigmaSquares()/
 *                                    GetLevels()  include/ORBextractor.h:63-83
 *   pgorb_descriptor_distance        static ORBmatcher::DescriptorDistance(a, b)
 *                                    include/ORBmatcher.h:44, src/ORBmatcher.cc:1651-1667
 *   pgorb_hamming_*                  the candidate-scan inner loops of the matchers
 *                                    (src/ORBmatcher.cc:438-459 etc.) as an all-pairs
 *                                    superset
 *
 * Conventions: plain pointers and sizes, caller owns every buffer, integer
 * status codes (0 = ok, <0 = error, text via pgorb_last_error) instead of the
 * reference's assert()/silent return (src/ORBextractor.cc:1045-1049).  One
 * context per host thread and device, like the reference's non-re-entrant
 * extractor.  There is NO CPU fallback: without a HIP device pgorb_create
 * fails with PGORB_E_NODEVICE.
 *
 * "_device" entry points take device pointers and a hipStream_t (passed as
 * void*) and are asynchronous on that stream; the others take host pointers
 * and are synchronous.
 */
#ifndef PGORB_H
#define PGORB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PGORB_MAX_LEVELS 16

enum {
    PGORB_OK = 0,
    PGORB_E_ARG = -1,        /* bad argument                                            */
    PGORB_E_TOOSMALL = -2,   /* a pyramid level has no 30-px cell (the reference divides
                                by zero there) or aspect < 0.5 (DistributeOctTree nIni=0) */
    PGORB_E_CAP = -3,        /* output capacity too small; nothing truncated silently   */
    PGORB_E_NODEVICE = -4,   /* no HIP device / wrong architecture                      */
    PGORB_E_HIP = -5,        /* HIP runtime error (see pgorb_last_error)                */
    PGORB_E_LIMIT = -6,      /* exceeds max_width/max_height/max_batch of the context, or a
                              * level larger than 4095 px a side                          */
    PGORB_E_OVERFLOW = -7    /* internal candidate capacity exceeded                    */
};

typedef struct pgorb_ctx pgorb_ctx;

typedef struct pgorb_params {
    int32_t nfeatures;       /* ORBextractor_nFeatures  (src/Tracking.cc:131)            */
    float   scale_factor;    /* ORBextractor_scaleFactor                                 */
    int32_t nlevels;         /* ORBextractor_nLevels, 1..PGORB_MAX_LEVELS                */
    int32_t ini_th_fast;     /* ORBextractor_iniThFAST                                   */
    int32_t min_th_fast;     /* ORBextractor_minThFAST                                   */
    int32_t max_width;       /* largest frame the context must hold (<= 4095)            */
    int32_t max_height;
    int32_t max_batch;       /* frames per pgorb_extract_batch* call                     */
    int32_t device;          /* HIP device ordinal                                       */
    int32_t blur_tie_mode;   /* 0: OpenCV x86-64 binary rounding of the blur column pass
                                (ties to even for x < (w & ~3)); 1: half-up everywhere   */
} pgorb_params;

/* == cv::KeyPoint of OpenCV 2.4 (pt.x, pt.y, size, angle, response, octave, class_id) */
typedef struct pgorb_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} pgorb_keypoint;

int  pgorb_create(const pgorb_params* params, pgorb_ctx** out);
void pgorb_destroy(pgorb_ctx* ctx);
/* Last error text of `ctx` (or of the failed pgorb_create when ctx == NULL). */
const char* pgorb_last_error(const pgorb_ctx* ctx);

int  pgorb_levels(const pgorb_ctx* ctx);
/* nlevels+1 entries each (the reference sizes its tables nlevels+1, ORBextractor.cc:415-433);
 * any pointer may be NULL. */
int  pgorb_scale_tables(const pgorb_ctx* ctx, float* scale, float* inv_scale,
                        float* sigma2, float* inv_sigma2);
/* nlevels+1 entries (last = the fork's unused remainder slot, ORBextractor.cc:446) */
int  pgorb_features_per_level(const pgorb_ctx* ctx, int32_t* out);
/* Keypoint capacity that a w x h frame can never exceed: sum over levels of
 * max(quota + 2, 4 * nIni) (DistributeOctTree may overshoot its quota by up to 2 when the
 * last split adds 3 nodes, ORBextractor.cc:730; nIni = initial nodes, :543).
 * <0 when the frame is unusable (PGORB_E_TOOSMALL / PGORB_E_LIMIT). */
int  pgorb_max_keypoints(const pgorb_ctx* ctx, int w, int h);

/* Page-locked host memory for frames / results: host-buffer entry points run at PCIe speed when
 * their buffers come from here (pageable memory is staged by the driver at a fraction of it). */
void* pgorb_host_alloc(int64_t bytes);
void  pgorb_host_free(void* p);
/* Page-lock memory the CALLER owns (a frame buffer a decoder reuses call after call, e.g. the cv::Mat behind
 * Frame.cc:251-257): pgorb_extract* then uploads it with ONE DMA instead of copying it through the library's staging
 * slot (the host thread's 2 MB memcpy per 1080p frame).  The caller keeps the obligation the ownership implies:
 * pgorb_host_unregister before the memory is freed or remapped -- which is why the library does not do this behind
 * the caller's back, keyed by address.  Returns PGORB_OK, PGORB_E_ARG, or PGORB_E_HIP (the range cannot be locked). */
int   pgorb_host_register(void* p, int64_t bytes);
int   pgorb_host_unregister(void* p);

/* One frame, host buffers.  gray: h rows of `stride` bytes.  kps[cap], desc[cap*32]. */
int  pgorb_extract(pgorb_ctx* ctx, const uint8_t* gray, int w, int h, int stride,
                   pgorb_keypoint* kps, uint8_t* desc, int cap, int* n);
/* nframes frames of equal size, host buffers; frame f writes kps[f*cap_per_frame ...],
 * desc[(f*cap_per_frame)*32 ...], n[f]. */
int  pgorb_extract_batch(pgorb_ctx* ctx, const uint8_t* const* gray, int nframes,
                         int w, int h, int stride,
                         pgorb_keypoint* kps, uint8_t* desc, int cap_per_frame, int* n);
/* Same, everything resident in HBM: d_gray = nframes planes `frame_stride` bytes apart.
 * Asynchronous on `hip_stream`.  d_n[f] receives the true count even when it exceeds
 * cap_per_frame (then only the first cap_per_frame entries were written). */
int  pgorb_extract_batch_device(pgorb_ctx* ctx, const uint8_t* d_gray, int nframes,
                                int w, int h, int stride, int64_t frame_stride,
                                pgorb_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame,
                                int32_t* d_n, void* hip_stream);
/* Colour ingest: Tracking::GrabImageMonocular's cvtColor (src/Tracking.cc:247-260) fused in
 * front of the extractor.  d_img = nframes interleaved 8-bit images with `channels` = 3 or 4
 * bytes per pixel, rgb_order != 0 for R,G,B(,A) (Camera_RGB: 1), 0 for B,G,R(,A).
 * gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14 (OpenCV 2.4 CV_RGB2GRAY, 8U).  The grey
 * planes are written straight into the context's pyramid level 0. */
int  pgorb_extract_batch_color_device(pgorb_ctx* ctx, const uint8_t* d_img, int nframes,
                                      int w, int h, int stride, int64_t frame_stride,
                                      int channels, int rgb_order,
                                      pgorb_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame,
                                      int32_t* d_n, void* hip_stream);
/* The reader's geometry as well: frames exactly as decoded, upright frames into the extractor.
 * rotate_degrees in {0, 90, 180, 270} as the video metadata says (src/io/image_sequence_reader.cc:
 * 186-205: 90 = cv::flip(raw.t(), 0), 180 = cv::flip(raw, -1), 270 = cv::flip(raw.t(), 1)), then
 * the optional --vertical_flip / --horizontal_flip of the wrapper source (:53-58, :212-222;
 * src/optical_trajectories.cc:49-52,84-85), then Tracking's grey conversion as above
 * (channels = 1: already grey).  Extracted frames are src_h x src_w for 90 / 270. */
int  pgorb_extract_batch_ingest_device(pgorb_ctx* ctx, const uint8_t* d_img, int nframes,
                                       int src_w, int src_h, int stride, int64_t frame_stride,
                                       int channels, int rgb_order, int rotate_degrees,
                                       int vertical_flip, int horizontal_flip,
                                       pgorb_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame,
                                       int32_t* d_n, void* hip_stream);

/* Device status word of the last *_device call: 0 or PGORB_E_OVERFLOW.  Synchronises. */
int  pgorb_check_async(pgorb_ctx* ctx, void* hip_stream);

/* ORBmatcher::DescriptorDistance on two 32-byte descriptors (host, no device work). */
int  pgorb_descriptor_distance(const uint8_t* a, const uint8_t* b);
/* All-pairs Hamming: out[i*nb + j] = distance(a[i], b[j]).  Host buffers. */
int  pgorb_hamming_matrix(pgorb_ctx* ctx, const uint8_t* a, int na, const uint8_t* b, int nb,
                          uint16_t* out);
/* Best and second-best train descriptor per query (first minimum wins, strict <, like
 * the bestDist/bestDist2 scans in src/ORBmatcher.cc:438-459).  best_idx = -1 and
 * best = second = 65535 when nb == 0; second = 65535 when nb == 1. */
int  pgorb_hamming_best2(pgorb_ctx* ctx, const uint8_t* a, int na, const uint8_t* b, int nb,
                         int32_t* best_idx, uint16_t* best, uint16_t* second);
/* Batched, resident: pair p matches descriptors of frame qa[p] (queries) against frame
 * qb[p] (train) inside one extract batch layout (desc base + f*cap_per_frame*32, counts
 * d_n[f] clamped to cap_per_frame).  Outputs are [npairs][cap_per_frame]. */
int  pgorb_match_batch_device(pgorb_ctx* ctx, const uint8_t* d_desc, const int32_t* d_n,
                              int cap_per_frame, const int32_t* d_pair_query,
                              const int32_t* d_pair_train, int npairs,
                              int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second,
                              void* hip_stream);

/* ---- Frame grid + guided matcher for initialisation ---------------------------------------
 *   pgorb_frame_grid*            Frame::AssignFeaturesToGrid / PosInGrid
 *                                thirdparty/orb-slam2/src/Frame.cc:234-249, 386-396 (64 x 48 grid,
 *                                include/Frame.h:37-38) as CSR: cell = col*48 + row,
 *                                start[3073], idx[] in insertion (= keypoint) order
 *   pgorb_search_for_initialization*   ORBmatcher::SearchForInitialization(F1, F2,
 *                                vbPrevMatched, vnMatches12, windowSize)
 *                                src/ORBmatcher.cc:407-522 incl. GetFeaturesInArea
 *                                (Frame.cc:331-384), TH_LOW = 50, the nnratio test, the
 *                                30-bin rotation histogram and ComputeThreeMaxima (:1605-1646)
 * Keypoints are taken as already undistorted (pilotguru calibrations with k1 == 0 skip
 * cv::undistortPoints, Frame.cc:410-414) and the bounds are the image rectangle
 * (Frame.cc:461-466): pass min_x = 0, max_x = cols, min_y = 0, max_y = rows.
 * The batched device forms work on the layout of pgorb_extract_batch_device. */
/*   pgorb_undistort_keypoints*   Frame::UndistortKeyPoints (src/Frame.cc:408-438) =
 *       cv::undistortPoints(pts, pts, mK, mDistCoef, Mat(), mK): camera = {fx, fy, cx, cy},
 *       dist = {k1, k2, p1, p2, k3} (as float, like mK / mDistCoef).  k1 == 0 copies the
 *       keypoints unchanged (:410-414).  Only pt.x / pt.y change.
 *   pgorb_image_bounds           Frame::ComputeImageBounds (src/Frame.cc:440-467): bounds =
 *       {mnMinX, mnMaxX, mnMinY, mnMaxY} from the four undistorted image corners. */
int  pgorb_undistort_keypoints(pgorb_ctx* ctx, const pgorb_keypoint* kps, int n,
                               const float camera[4], const float dist[5], pgorb_keypoint* out);
int  pgorb_undistort_keypoints_batch_device(pgorb_ctx* ctx, const pgorb_keypoint* d_kps,
                               const int32_t* d_n, int nframes, int cap_per_frame,
                               const float camera[4], const float dist[5],
                               pgorb_keypoint* d_out, void* hip_stream);
int  pgorb_image_bounds(int cols, int rows, const float camera[4], const float dist[5], float bounds[4]);
#define PGORB_GRID_COLS 64
#define PGORB_GRID_ROWS 48
#define PGORB_GRID_CELLS (PGORB_GRID_COLS * PGORB_GRID_ROWS)
int  pgorb_frame_grid(pgorb_ctx* ctx, const pgorb_keypoint* kps, int n,
                      float min_x, float max_x, float min_y, float max_y,
                      int32_t* grid_start /*[3073]*/, int32_t* grid_idx /*[n]*/);
int  pgorb_frame_grid_batch_device(pgorb_ctx* ctx, const pgorb_keypoint* d_kps, const int32_t* d_n,
                                   int nframes, int cap_per_frame,
                                   float min_x, float max_x, float min_y, float max_y,
                                   int32_t* d_grid_start /*[nframes][3073]*/,
                                   int32_t* d_grid_idx /*[nframes][cap]*/, void* hip_stream);
/* prev_matched: float[2*n1] in/out (vbPrevMatched); matches12: int32[n1] out (-1 = none);
 * returns nmatches (>= 0) or an error code. */
int  pgorb_search_for_initialization(pgorb_ctx* ctx,
                                     const pgorb_keypoint* kps1, const uint8_t* desc1, int n1,
                                     const pgorb_keypoint* kps2, const uint8_t* desc2, int n2,
                                     float min_x, float max_x, float min_y, float max_y,
                                     float* prev_matched, int32_t* matches12,
                                     int window_size, float nnratio, int check_orientation);
/* pair p: F1 = frame pair_f1[p], F2 = frame pair_f2[p]; d_prev_matched / d_matches12 are
 * [npairs][cap]; d_nmatches[npairs]. */
int  pgorb_search_for_initialization_batch_device(pgorb_ctx* ctx, const pgorb_keypoint* d_kps,
                                     const uint8_t* d_desc, const int32_t* d_n, int cap_per_frame,
                                     const int32_t* d_grid_start, const int32_t* d_grid_idx,
                                     const int32_t* d_pair_f1, const int32_t* d_pair_f2, int npairs,
                                     float min_x, float max_x, float min_y, float max_y,
                                     float* d_prev_matched, int32_t* d_matches12, int32_t* d_nmatches,
                                     int window_size, float nnratio, int check_orientation,
                                     void* hip_stream);

/* ---- Guided matchers of the tracking thread (next tier of SURVEY.md 8a row a11) -------------
 *   pgorb_search_by_projection_points   ORBmatcher::SearchByProjection(Frame &F, const
 *       vector<MapPoint*> &vpMapPoints, float th)  src/ORBmatcher.cc:46-131 incl.
 *       RadiusByViewingCos :133-139 -- the local-map matcher of Tracking::SearchLocalPoints
 *       (src/Tracking.cc:1134-1184).  Per map point the caller passes what MapPoint carries after
 *       Frame::isInFrustum: mbTrackInView && !isBad() (valid), mTrackProjX/Y, mnTrackScaleLevel,
 *       mTrackViewCos, GetDescriptor(), Observations() > 0.
 *   pgorb_search_by_projection_frame    the matching loop of ORBmatcher::SearchByProjection(Frame
 *       &CurrentFrame, const Frame &LastFrame, float th, bool bMono)  src/ORBmatcher.cc:1355-1474
 *       for bMono = true, given the projections (u, v) of the last frame's map points into the
 *       current frame (the cv::Mat pose arithmetic of :1343-1377 stays with the caller, who owns
 *       the poses): valid = has a map point && !outlier && invzc >= 0 && inside the image
 *       bounds; octave/angle of the last frame's keypoint; best match only, TH_HIGH, rotation
 *       histogram + ComputeThreeMaxima.
 * Common state: kp_has_point[i] != 0 when the frame's keypoint i already holds a map point with
 * Observations() > 0 before the call (:79-81 / :1397-1399); assigned[i] receives the index of the
 * query (map point) written to F.mvpMapPoints[i] by this call, or -1.  Queries are processed in
 * order (each assignment changes what later queries may take): the results equal the reference's
 * sequence; since round 4 the device decides provably independent queries together. */
int  pgorb_search_by_projection_points(pgorb_ctx* ctx,
        const pgorb_keypoint* kps, const uint8_t* desc, int n,            /* the frame F             */
        float min_x, float max_x, float min_y, float max_y,
        const uint8_t* kp_has_point,                                      /* [n]                     */
        int npoints, const uint8_t* valid, const float* proj_x, const float* proj_y,
        const int32_t* level, const float* view_cos, const uint8_t* point_desc,
        const uint8_t* point_has_obs, float th, float nnratio,
        int32_t* assigned /*[n]*/);                                       /* returns nmatches        */
int  pgorb_search_by_projection_frame(pgorb_ctx* ctx,
        const pgorb_keypoint* kps, const uint8_t* desc, int n,            /* CurrentFrame            */
        float min_x, float max_x, float min_y, float max_y,
        const uint8_t* kp_has_point,
        int nlast, const uint8_t* valid, const float* u, const float* v,
        const int32_t* last_octave, const float* last_angle, const uint8_t* point_desc,
        const uint8_t* point_has_obs, float th, int check_orientation,
        int32_t* assigned /*[n]*/);                                       /* returns nmatches        */

/*   pgorb_search_by_projection_keyframe   the matching loop of ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF,
 *       const set<MapPoint*> &sAlreadyFound, float th, int ORBdist)  src/ORBmatcher.cc:1476-1603 -- the projection search of
 *       Tracking::Relocalization (src/Tracking.cc:1434: th 10, ORBdist 100; :1448: th 3, ORBdist 64).  Per map point i of the
 *       key frame the caller passes what its pose arithmetic (:1497-1517, cv::Mat) produced: valid = pMP && !pMP->isBad();
 *       already_found = sAlreadyFound.count(pMP); (u, v) the projection into the current frame; dist3d = |x3Dw - Ow|;
 *       min / max_distance = pMP->GetMin/MaxDistanceInvariance(); kf_angle = pKF->mvKeysUn[i].angle; GetDescriptor().
 *       The device does the rest: the image-bounds and depth-range tests (:1512-1526), MapPoint::PredictScale
 *       (src/MapPoint.cc:516-531) with log_scale_factor = CurrentFrame.mfLogScaleFactor, radius th * mvScaleFactors[level],
 *       levels level - 1 .. level + 1, best match only with ORBdist, rotation histogram.  Here ANY point in
 *       CurrentFrame.mvpMapPoints[i2] blocks a keypoint (:1542-1543: no Observations() test): kp_has_point[i2] != 0.
 *       PredictScale's `log` is the platform's logf in the reference; here a fixed double-precision sequence rounded once
 *       (pgorb_log_f; DESIGN.md section 5, parity contract 5), and pgorb_log_scale_factor is mfLogScaleFactor under it. */
int  pgorb_search_by_projection_keyframe(pgorb_ctx* ctx,
        const pgorb_keypoint* kps, const uint8_t* desc, int n,            /* CurrentFrame            */
        float min_x, float max_x, float min_y, float max_y,
        const uint8_t* kp_has_point,
        int npoints, const uint8_t* valid, const uint8_t* already_found, const float* u, const float* v,
        const float* dist3d, const float* min_distance, const float* max_distance,
        const float* kf_angle, const uint8_t* point_desc,
        float log_scale_factor, float th, int orb_dist, int check_orientation,
        int32_t* assigned /*[n]*/);                                       /* returns nmatches        */
float pgorb_log_f(float x);
float pgorb_log_scale_factor(const pgorb_ctx* ctx);                      /* Frame::mfLogScaleFactor (Frame.cc:188)     */
int   pgorb_predict_scale(const pgorb_ctx* ctx, float max_distance, float current_dist);   /* MapPoint::PredictScale */

/*   pgorb_search_by_bow   ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*>
 *       &vpMapPointMatches)  src/ORBmatcher.cc:161-290 (Tracking::TrackReferenceKeyFrame,
 *       src/Tracking.cc:758; relocalisation :1359).  Both FeatureVectors come as the CSR arrays
 *       pgorb_bow_vectors produces (ascending node ids).  kf_point_valid[i] = the key frame's
 *       keypoint i has a map point that is not bad.  matches[j] = key-frame keypoint index whose
 *       map point was written to vpMapPointMatches[j], or -1.  Returns nmatches. */
int  pgorb_search_by_bow(pgorb_ctx* ctx,
        const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_point_valid, int nkf,
        const uint32_t* kf_fv_node, const int32_t* kf_fv_start, const uint32_t* kf_fv_feat, int kf_nfv,
        const uint8_t* f_desc, const float* f_angle, int nf,
        const uint32_t* f_fv_node, const int32_t* f_fv_start, const uint32_t* f_fv_feat, int f_nfv,
        float nnratio, int check_orientation, int32_t* matches /*[nf]*/);

/* Batched, resident forms of the three matchers above (round 3; the single calls are one-pair batches of these).
 * Frames live in the layout of pgorb_extract_batch_device (keypoints / descriptors `cap_per_frame` apart, counts d_n),
 * grids as pgorb_frame_grid_batch_device writes them; pair p matches its queries against frame d_pair_frame[p]
 * (NULL: frame p).  Query arrays are [npairs][qcap] with d_nq[p] entries in use; d_kp_has_point (NULL = none) and
 * d_assigned are [npairs][cap_per_frame], d_nmatches [npairs].  The reference's order dependence (an assignment is seen
 * by every later query, ORBmatcher.cc:75-79, :1398-1402, :236-240) stays inside a pair; pairs run concurrently.  SearchByProjection:
 * two passes (candidates and distances of every query in parallel; then one workgroup per pair decides the queries in rounds
 * of independent ones, every decision equal to the reference's sequence -- round 4);
 * SearchByBoW: one wave per common vocabulary node (a frame feature belongs to one node, so nodes are independent).  Call sites: Tracking::SearchLocalPoints (Tracking.cc:1175), TrackWithMotionModel (:876, :882),
 * TrackReferenceKeyFrame (:758). */
int  pgorb_search_by_projection_points_batch_device(pgorb_ctx* ctx,
        const pgorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int cap_per_frame,
        const int32_t* d_grid_start, const int32_t* d_grid_idx, const int32_t* d_pair_frame, int npairs,
        float min_x, float max_x, float min_y, float max_y, const uint8_t* d_kp_has_point,
        int qcap, const int32_t* d_nq, const uint8_t* d_valid, const float* d_proj_x, const float* d_proj_y,
        const int32_t* d_level, const float* d_view_cos, const uint8_t* d_point_desc, const uint8_t* d_point_has_obs,
        float th, float nnratio, int32_t* d_assigned, int32_t* d_nmatches, void* hip_stream);
int  pgorb_search_by_projection_frame_batch_device(pgorb_ctx* ctx,
        const pgorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int cap_per_frame,
        const int32_t* d_grid_start, const int32_t* d_grid_idx, const int32_t* d_pair_frame, int npairs,
        float min_x, float max_x, float min_y, float max_y, const uint8_t* d_kp_has_point,
        int qcap, const int32_t* d_nq, const uint8_t* d_valid, const float* d_u, const float* d_v,
        const int32_t* d_last_octave, const float* d_last_angle, const uint8_t* d_point_desc, const uint8_t* d_point_has_obs,
        float th, int check_orientation, int32_t* d_assigned, int32_t* d_nmatches, void* hip_stream);
int  pgorb_search_by_projection_keyframe_batch_device(pgorb_ctx* ctx,
        const pgorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int cap_per_frame,
        const int32_t* d_grid_start, const int32_t* d_grid_idx, const int32_t* d_pair_frame, int npairs,
        float min_x, float max_x, float min_y, float max_y, const uint8_t* d_kp_has_point,
        int qcap, const int32_t* d_nq, const uint8_t* d_valid, const uint8_t* d_already_found, const float* d_u, const float* d_v,
        const float* d_dist3d, const float* d_min_distance, const float* d_max_distance, const float* d_kf_angle,
        const uint8_t* d_point_desc, float log_scale_factor, float th, int orb_dist, int check_orientation,
        int32_t* d_assigned, int32_t* d_nmatches, void* hip_stream);
/* FeatureVector of every frame on the device: from the per-feature node ids pgorb_bow_transform_device wrote
 * (d_node [nframes][cap]) the CSR arrays of DBoW2's FeatureVector (FeatureVector.cpp:31-45: node ids ascending,
 * feature indices of a node in feature order): d_fv_node / d_fv_feat [nframes][cap], d_fv_start [nframes][cap + 1],
 * d_nfv [nframes].  Equal to what pgorb_bow_vectors returns as fv_node / fv_start / fv_feat. */
int  pgorb_feature_vectors_batch_device(pgorb_ctx* ctx, const uint32_t* d_node, const int32_t* d_n, int nframes, int cap_per_frame,
        uint32_t* d_fv_node, int32_t* d_fv_start, uint32_t* d_fv_feat, int32_t* d_nfv, void* hip_stream);
/* pair p: key frame d_pair_kf[p], frame d_pair_f[p] of the same batch; d_kf_point_valid and d_matches [npairs][cap]. */
int  pgorb_search_by_bow_batch_device(pgorb_ctx* ctx,
        const pgorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int cap_per_frame,
        const uint32_t* d_fv_node, const int32_t* d_fv_start, const uint32_t* d_fv_feat, const int32_t* d_nfv,
        const int32_t* d_pair_kf, const int32_t* d_pair_f, int npairs, const uint8_t* d_kf_point_valid,
        float nnratio, int check_orientation, int32_t* d_matches, int32_t* d_nmatches, void* hip_stream);

/* ---- ORB vocabulary (DBoW2 TemplatedVocabulary<FORB::TDescriptor, FORB>) -----------------
 *   pgorb_vocab_load_text     ORBVocabulary(text_file) -> TemplatedVocabulary::loadFromTextFile
 *                             thirdparty/orb-slam2/src/ORBVocabulary.cc:7-9,
 *                             thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1337-1420
 *   pgorb_bow_transform*      TemplatedVocabulary::transform(feature, word_id, weight, nid,
 *                             levelsup)  TemplatedVocabulary.h:1217-1259 for every feature
 *   pgorb_bow_vectors         the accumulation part of transform(features, BowVector&,
 *                             FeatureVector&, levelsup)  TemplatedVocabulary.h:1126-1194 with
 *                             BowVector::addWeight/normalize (BowVector.cpp:34-84) and
 *                             FeatureVector::addFeature (FeatureVector.cpp:31-45), called from
 *                             Frame::ComputeBoW (thirdparty/orb-slam2/src/Frame.cc:399-406)
 * A vocabulary is a flat little-endian blob (layout in pilotguru_amd/vocab.py and bow.hip) so
 * that one rank can parse it and broadcast it to its peers with a single RCCL broadcast.
 * Deviation from the reference loader: empty lines are skipped (the reference appends a bogus
 * root child with an uninitialised descriptor for a trailing newline). */
typedef struct pgorb_vocab pgorb_vocab;
int  pgorb_vocab_load_text(const char* path, pgorb_vocab** out);
/* As pgorb_vocab_load_text, through a binary cache `<path>.pgvoc` beside the text file (header: the text file's size and
 * modification time; then the blob): a matching cache is loaded instead of parsing the text, anything else parses and rewrites
 * the cache (failures to write are ignored).  *from_cache (may be NULL): 1 when the cache was used. */
int  pgorb_vocab_load_cached(const char* path, pgorb_vocab** out, int* from_cache);
int  pgorb_vocab_from_blob(const void* blob, int64_t nbytes, pgorb_vocab** out);   /* copies */
int  pgorb_vocab_blob(const pgorb_vocab* v, const void** blob, int64_t* nbytes);
int  pgorb_vocab_info(const pgorb_vocab* v, int* k, int* L, int* nnodes, int* nwords,
                      int* scoring, int* weighting);
void pgorb_vocab_free(pgorb_vocab* v);
/* Make a vocabulary resident on the context's GPU (host blob: H2D copy; device blob, e.g. the
 * receive buffer of the broadcast: D2D copy on `hip_stream`). */
int  pgorb_vocab_upload(pgorb_ctx* ctx, const pgorb_vocab* v);
int  pgorb_vocab_upload_device(pgorb_ctx* ctx, const void* d_blob, int64_t nbytes, void* hip_stream);
/* ---- The vocabulary on every GPU of the node: ONE RCCL broadcast over xGMI --------------------------------------------
 * Reference: the single process loads the vocabulary once and shares it with every System by pointer
 * (src/optical_trajectories.cc:87-94, thirdparty/orb-slam2/src/ORBVocabulary.cc:7-9).  With one extractor context per
 * GPU the pointer share becomes one ncclBroadcast of the flat blob, issued from librccl directly (csrc/comm.hip; the
 * library is opened on first use).  It is the only collective of the path: frames / rides are sharded, nothing else
 * is exchanged.
 *   pgorb_comm_create_local   ONE process, one host thread per device (optical_trajectories --devices=0,1,...): the
 *                             group's ranks are the DISTINCT devices of ctxs[0..nctx) (ncclCommInitAll); contexts
 *                             that share a device share its rank and receive by a device-to-device copy.
 *   pgorb_comm_unique_id /    one PROCESS per GPU (bench.py under torch.distributed.run): rank 0 makes the 128-byte
 *   pgorb_comm_create_rank    id, the launcher's control plane hands it to the other ranks, each calls create_rank
 *                             (ncclCommInitRank; collective over the ranks).
 *   pgorb_vocab_broadcast     local form: `root` indexes ctxs, `v` = the parsed vocabulary.  Rank form: `root` is a
 *                             rank, `v` is read on that rank only (NULL elsewhere; the byte count travels first).
 *                             The receive buffer is the context's own vocabulary arena; every receiver validates
 *                             the blob's structure on its device before the vocabulary counts as resident.
 *                             *seconds (may be NULL): wall time of the collective alone.
 *   pgorb_comm_library        which librccl the library resolved (the file ncclGetUniqueId lives in).  The library asks
 *                             the loader for a librccl the PROCESS already holds first (RTLD_NOLOAD: under
 *                             torch.distributed that is torch's bundled copy) and loads one by name only when there is
 *                             none, so that one process never runs two RCCLs.  *preloaded (may be NULL) = 1 when it was
 *                             already mapped.  Returns 0, or PGORB_E_HIP with the loader's message in `path`. */
#define PGORB_COMM_ID_BYTES 128
typedef struct pgorb_comm pgorb_comm;
int  pgorb_comm_library(char* path, int cap, int* preloaded);
int  pgorb_comm_unique_id(void* id /*[PGORB_COMM_ID_BYTES]*/);
int  pgorb_comm_create_local(pgorb_ctx* const* ctxs, int nctx, pgorb_comm** out);
int  pgorb_comm_create_rank(pgorb_ctx* ctx, int rank, int nranks, const void* id, pgorb_comm** out);
int  pgorb_comm_ranks(const pgorb_comm* comm);
int  pgorb_vocab_broadcast(pgorb_comm* comm, int root, const pgorb_vocab* v, double* seconds);
void pgorb_comm_destroy(pgorb_comm* comm);
/* Per-feature word id, word weight and the ancestor node at level L - levelsup (0 = root).
 * Host buffers / device buffers + stream. */
int  pgorb_bow_transform(pgorb_ctx* ctx, const uint8_t* desc, int n, int levelsup,
                         uint32_t* word, double* weight, uint32_t* node);
int  pgorb_bow_transform_device(pgorb_ctx* ctx, const uint8_t* d_desc, int n, int levelsup,
                                uint32_t* d_word, double* d_weight, uint32_t* d_node,
                                void* hip_stream);
/* Host: BowVector (ascending word ids, L1-normalised when the scoring type says so) and
 * FeatureVector (ascending node ids, feature indices in feature order) from the per-feature
 * results.  bow_* need n entries, fv_node/fv_start n+1 entries, fv_feat n entries.
 * scoring: 0 = L1_NORM ... (BowVector.h:36-53); weighting 0 = TF_IDF, 1 = TF. */
int  pgorb_bow_vectors(int n, const uint32_t* word, const double* weight, const uint32_t* node,
                       int scoring, int weighting,
                       uint32_t* bow_id, double* bow_val, int* n_bow,
                       uint32_t* fv_node, int32_t* fv_start, uint32_t* fv_feat, int* n_fv);
/* L1Scoring::score(v1, v2), ScoringObject.cpp:23-60 (host). */
double pgorb_bow_score_l1(const uint32_t* id1, const double* val1, int n1,
                          const uint32_t* id2, const double* val2, int n2);

/* ---- after the path: what TrackImageSequence does to the finished trajectory ----
 * Host functions, double precision, no context needed (the reference runs them once per segment on
 * the CPU).  Quaternions are (w, x, y, z) like the JSON; all arrays are row-major.
 *   pgorb_smooth_heading_directions  SmoothHeadingDirections(trajectory, sigma)  src/slam/smoothing.cc:11-47
 *                                    (cv::getGaussianKernel(4*sigma+1, sigma), cv::sepFilter2D with
 *                                    BORDER_REPLICATE along the trajectory, renormalisation); sigma <= 0 is the
 *                                    reference's CHECK failure -> PGORB_E_ARG
 *   pgorb_smooth_time_series         SmoothTimeSeries(values, timestamps, targets, sigma)  smoothing.cc:57-97
 *   pgorb_trajectory_pca             TrajectoryToPCA  src/slam/track_image_sequence.cc:16-29: cv::PCA with
 *                                    CV_PCA_DATA_AS_COL over the translations; eigenvectors[3][3] (rows, by
 *                                    descending eigenvalue), eigenvalues[3], mean[3] (may be NULL).  The plane of
 *                                    :92 is the first two rows; :83-90 drops the trajectory when
 *                                    eigenvalues[2] > eigenvalues[1] * 1e-2.  n < 3 -> PGORB_E_LIMIT
 *   pgorb_project_directions         ProjectDirections  src/slam/horizontal_flatten.cc:7-30 -> dirs[n][2]
 *   pgorb_project_translations       ProjectTranslations  horizontal_flatten.cc:32-43 (in place)
 *   pgorb_turn_angles                Projected2DDirectionsToTurnAngles  horizontal_flatten.cc:45-63 */
int  pgorb_smooth_heading_directions(double* quat_wxyz /* [n][4] in/out */, int n, int sigma);
int  pgorb_smooth_time_series(const double* values, const double* times, int n,
                              const double* targets, int m, double sigma, double* out /* [m] */);
int  pgorb_trajectory_pca(const double* translations /* [n][3] */, int n,
                          double* eigenvectors /* [3][3] */, double* eigenvalues /* [3] */, double* mean /* [3] or NULL */);
int  pgorb_project_directions(const double* quat_wxyz, int n, const double* plane /* [2][3] */, double* dirs /* [n][2] */);
int  pgorb_project_translations(double* translations /* [n][3] in/out */, int n, const double* plane);
int  pgorb_turn_angles(const double* dirs /* [n][2] */, int n, double* turn /* [n] */);

/* ---- fit_motion's velocity calibration (BASELINE configs[4]), K9 calib.hip ----
 * Inputs are the three recorder series as arrays: GPS speed [n_gps] + time, gyroscope rates
 * [n_rot][3] (rad/s) + time, accelerations [n_acc][3] + time; times in microseconds, increasing.
 *   pgorb_fit_num_windows        number of sliding windows = ceil(n_gps / locations_shift_step)
 *                                (the loop of src/fit_motion.cc:173-176)
 *   pgorb_fit_velocity_windows   for every window: AccelerometerCalibrator over the window's GPS fixes
 *                                (src/calibration/velocity.cc:30-180) minimised from x = 0 with
 *                                LBFGSpp::LBFGSSolver (thirdparty/LBFGS/LBFGS.h:78-181, epsilon 1e-5,
 *                                max_iterations = optimization_iters; fit_motion.cc:163-190).  All windows
 *                                run concurrently on the GPU, one lane each.  x[w][9] = global bias, local
 *                                bias, initial velocity; residual[w] = loss; niter[w] = iterations, or -2 / -3
 *                                where LBFGSpp would throw (line-search step below 1e-20 / above 1e20).
 *   pgorb_calibrator_eval        AccelerometerCalibrator::operator() (velocity.cc:182-194) of ONE calibrator
 *                                at n_points parameter vectors xin[n_points][9] -> fx[n_points], grad[n_points][9]
 *   pgorb_fit_motion_velocities  ComputeAndSaveForwardVelocitiesFromImu (fit_motion.cc:151-290) up to the JSON:
 *                                window fits on the GPU, then IntegrateTrajectory, per-sample averaging,
 *                                SmoothTimeSeries and the forward axis on the host.  out_* need n_rot + n_acc entries. */
/*   pgorb_principal_rotation_axes         GetPrincipalRotationAxes (src/calibration/rotation.cc:16-57): gyroscope
 *                                         rates integrated over intervals of integration_interval_usec, cv::PCA over
 *                                         the quaternion vector parts; eigenvectors[3][3] rows by descending
 *                                         eigenvalue, row 0 = the vehicle's vertical axis (fit_motion.cc:322-329).
 *                                         Fewer than 3 integrated intervals -> PGORB_E_LIMIT.  Host.
 *   pgorb_angular_velocities_around_axis  GetAngularVelocitiesAroundAxisDirect (rotation.cc:111-129) = the steering
 *                                         output of fit_motion (fit_motion.cc:130-148); axis must have norm 1 +- 1e-2. Host. */
int  pgorb_principal_rotation_axes(const double* rotations /* [n][3] */, const int64_t* rot_time_usec, int n,
                                   int64_t integration_interval_usec, double* eigenvectors /* [3][3] */);
int  pgorb_angular_velocities_around_axis(const double* rotations, int n, const double* axis /* [3] */, double* out /* [n] */);
/*   pgorb_kahan_sum                       KahanSum<T>::add over n vectors of dim components (include/math/math.hpp:8-25),
 *                                         the accumulator of fit_motion's forward-axis estimate.  Host. */
int  pgorb_kahan_sum(const double* values /* [n][dim] */, int n, int dim, double* sum /* [dim] */);
int  pgorb_fit_num_windows(int n_gps, int locations_shift_step);
int  pgorb_fit_velocity_windows(pgorb_ctx* ctx, const double* gps_velocity, const int64_t* gps_time_usec, int n_gps,
                                const double* rotations, const int64_t* rot_time_usec, int n_rot,
                                const double* accelerations, const int64_t* acc_time_usec, int n_acc,
                                int locations_batch_size, int locations_shift_step, int optimization_iters,
                                double* x /* [nw][9] */, double* residual /* [nw] */, int32_t* niter /* [nw] */);
int  pgorb_calibrator_eval(pgorb_ctx* ctx, const double* gps_velocity, const int64_t* gps_time_usec, int n_gps,
                           const double* rotations, const int64_t* rot_time_usec, int n_rot,
                           const double* accelerations, const int64_t* acc_time_usec, int n_acc,
                           const double* xin, int n_points, double* fx, double* grad);
int  pgorb_fit_motion_velocities(pgorb_ctx* ctx, const double* gps_velocity, const int64_t* gps_time_usec, int n_gps,
                                 const double* rotations, const int64_t* rot_time_usec, int n_rot,
                                 const double* accelerations, const int64_t* acc_time_usec, int n_acc,
                                 const double* vertical_axis /* [3] */, int locations_batch_size, int locations_shift_step,
                                 int optimization_iters, double post_smoothing_sigma_sec,
                                 double forward_axis_inference_min_velocity_m_s, double forward_axis_inference_min_rotation_rad,
                                 int64_t* out_time_usec, double* out_velocity, int* n_out, double* forward_axis /* [3] */);

/* Per-stage device timing with HIP events recorded on the launch stream around the kernel
 * groups of every *_device call: stage 0 = pyramid chain (K1, nlevels-1 launches), 1 = FAST
 * cells (K2), 2 = quadtree (K3), 3 = orientation+blur+rBRIEF (K4-6), 4 = Hamming match (K7).
 * pgorb_profile_begin arms up to max_calls calls (0 disarms); pgorb_profile_read synchronises,
 * writes the mean milliseconds per call of each stage into ms[5] and returns the number of
 * calls averaged. */
#define PGORB_NSTAGES 5
int  pgorb_profile_begin(pgorb_ctx* ctx, int max_calls);
int  pgorb_profile_read(pgorb_ctx* ctx, double* ms);

/* ---- Streamed ingest: frames that start in HOST memory --------------------------------------------------
 * Replaces the reference's frame loop around the extractor -- ImageSequenceSource::next() handing one decoded
 * frame at a time to System::TrackMonocular (src/io/image_sequence_reader.cc:138-208,
 * src/slam/track_image_sequence.cc:43-47) -- for callers whose frames are not already on the GPU.  A stream owns
 * `depth` (2..8) slots; a slot is one batch of up to `batch` w x h grey frames in PAGE-LOCKED host memory that
 * the decoder fills directly (pgorb_stream_input).  pgorb_stream_submit queues, without blocking: the upload of
 * the slot, K1..K6 on it, K7 of every frame against its predecessor (frame 0 against the last frame of the
 * previously submitted batch; after pgorb_stream_reset, or at the start, it has none and its best_idx are -1),
 * and the download of all results -- on three HIP streams, so the upload of batch i+1 and the download of
 * batch i-1 overlap the kernels of batch i.  pgorb_stream_wait blocks until the slot's results are in host
 * memory and returns the number of frames of the batch (< 0: error); the pointers stay valid until the slot is
 * submitted again: n[f] keypoints of frame f, kps[f * cap + i], desc[(f * cap + i) * 32], and for query
 * keypoint i of frame f best_idx / best / second [f * cap + i] exactly as pgorb_match_batch_device returns them.
 * One stream per context at a time; the context's other calls must not run between submit and wait. */
typedef struct pgorb_stream pgorb_stream;
int      pgorb_stream_create(pgorb_ctx* ctx, int w, int h, int batch, int depth, pgorb_stream** out);
/* The same stream for frames EXACTLY AS THE DECODER PRODUCES THEM: the reference's reader hands out RGB24 frames
 * (src/io/image_sequence_reader.cc:138-208, per-pixel copy :174-183), rotates them by the video metadata (:186-205),
 * the wrapper source flips them (:53-58, :212-222) and Tracking::GrabImageMonocular converts to grey
 * (thirdparty/orb-slam2/src/Tracking.cc:247-260) -- all on the host.  Here a slot holds src_w x src_h frames of
 * `channels` (1, 3, 4) interleaved bytes per pixel, row pitch src_w * channels, rgb_order as in
 * pgorb_extract_batch_ingest_device; rotation, flips and the grey conversion run on the device in front of K1 (one
 * pass: k_ingest / k_ingest_rows) and the extractor sees the upright (src_h x src_w for 90 / 270) grey frame.  Results,
 * matches across batch borders and the front-end stage are those of pgorb_stream_create on the host-converted frames. */
int      pgorb_stream_create_ingest(pgorb_ctx* ctx, int src_w, int src_h, int channels, int rgb_order, int rotate_degrees,
                                    int vertical_flip, int horizontal_flip, int batch, int depth, pgorb_stream** out);
void     pgorb_stream_destroy(pgorb_stream* s);                  /* pgorb_destroy(ctx) also destroys the context's live streams */
uint8_t* pgorb_stream_input(pgorb_stream* s, int slot);          /* batch * src_h * src_w * channels bytes, row pitch src_w * channels */
int      pgorb_stream_reset(pgorb_stream* s);                    /* the next batch starts a new ride */
int      pgorb_stream_submit(pgorb_stream* s, int slot, int nframes);
int      pgorb_stream_wait(pgorb_stream* s, int slot, const int32_t** n, const pgorb_keypoint** kps, const uint8_t** desc,
                           const int32_t** best_idx, const uint16_t** best, const uint16_t** second, int* cap);
/* The same stream for frames that are ALREADY RESIDENT on the device, with several batches in flight INSIDE the library
 * (round 5).  The reference's loop hands the extractor one frame after the other (Frame.cc:251-257); a caller with
 * resident frames submits batch after batch without blocking and the stream runs consecutive batches on `lanes`
 * (1..depth; 2 is what pays on an MI355X) independent extractor working sets -- lane 0 is the context itself, the others
 * are private siblings with the same parameters and options -- each on its own internal HIP stream, so that kernels of
 * neighbouring batches with different bottlenecks share the chip: K1 (HBM) beside K2 / K4-6 (VALU issue), K3 (latency),
 * K7 (matrix pipe).  Slot k runs on lane k % lanes.  Results are exactly those of the one-batch-at-a-time calls: K1..K6
 * of a batch depend on nothing outside it; what crosses batches (frame 0's match against the last frame of the
 * previously SUBMITTED batch, the front-end stage's state) is one section at the end of each batch's queue, chained in
 * submission order by an event.
 *   pgorb_stream_submit_device   d_frames: nframes grey planes (row pitch `stride`, `frame_stride` bytes apart), ready
 *                                where `hip_stream` (the caller's; NULL = the null stream) stands at the call; they
 *                                must stay valid until the slot's batch is complete (level 0 may alias them).
 *   pgorb_stream_wait_device     wait_on_host != 0: blocks until the slot's batch is complete and checks its status
 *                                word; otherwise makes `hip_stream` (NULL = the null stream, as in submit) wait for it
 *                                and returns at once.
 *                                DEVICE pointers, laid out as pgorb_stream_wait's, valid until the slot is submitted
 *                                again; with the front-end stage on, pgorb_stream_frontend_results hands out device
 *                                pointers as well.  Returns the number of frames of the batch. */
int      pgorb_stream_create_device(pgorb_ctx* ctx, int w, int h, int batch, int depth, int lanes, pgorb_stream** out);
int      pgorb_stream_submit_device(pgorb_stream* s, int slot, const uint8_t* d_frames, int nframes, int stride,
                                    int64_t frame_stride, void* hip_stream);
int      pgorb_stream_wait_device(pgorb_stream* s, int slot, int wait_on_host, void* hip_stream, const int32_t** d_n,
                                  const pgorb_keypoint** d_kps, const uint8_t** d_desc, const int32_t** d_best_idx,
                                  const uint16_t** d_best, const uint16_t** d_second, int* cap);
int      pgorb_stream_lanes(const pgorb_stream* s);
/* Optional front-end stage of the stream: what the reference's tracking thread does with every fresh Frame, run on
 * the device for the whole batch behind K7 -- Frame::AssignFeaturesToGrid with the given image bounds
 * (src/Frame.cc:234-249), ORBmatcher(nnratio, check_orientation).SearchForInitialization(previous frame, frame,
 * vbPrevMatched = the previous frame's keypoint positions, matches, window_size) as MonocularInitialization calls it
 * (Tracking.cc:583-597; frame 0 of a batch against the last frame of the previous one) and, when bow_levelsup >= 0
 * and a vocabulary is resident in the context, Frame::ComputeBoW's ORBVocabulary::transform (Frame.cc:399-406; per
 * feature word / weight / node, pgorb_bow_vectors turns them into BowVector / FeatureVector on the host).
 * Call with no batch in flight; the next batch starts a new ride.  After pgorb_stream_wait(slot):
 * pgorb_stream_frontend_results -> matches12[f * cap + i1] (-1 = none) and nmatches[f] for the pair (frame f - 1,
 * frame f) -- the first frame of a ride has no predecessor: it reports 0 matches and its matches12 row is undefined --, word / weight / node
 * [f * cap + i] (nullptr without BoW); valid until the slot is submitted again. */
int      pgorb_stream_frontend(pgorb_stream* s, float min_x, float max_x, float min_y, float max_y, int window_size,
                               float nnratio, int check_orientation, int bow_levelsup);
int      pgorb_stream_frontend_results(pgorb_stream* s, int slot, const int32_t** matches12, const int32_t** nmatches,
                                       const uint32_t** word, const double** weight, const uint32_t** node);

/* Measurement switches (no counterpart in the reference; results never depend on them -- the parity suite
 * runs under each).  Every option belongs to the CONTEXT it is set on (round 4: "matcher" / "match_mode" were process-wide
 * statics); the environment (PGORB_MATCH_POPCOUNT, PGORB_MATCH_MODE) only seeds pgorb_create.  ctx == NULL or an unknown
 * key -> PGORB_E_ARG.
 * key "matcher": 0 = the default (fp4 block-scaled MFMA for < 8192 descriptors per frame, unless PGORB_MATCH_POPCOUNT is
 * set), 1 = the ballot / popcount kernels BASELINE.json's north star describes, for every size.
 * key "match_mode": how the MFMA matcher gets its train descriptors -- -1 = chosen by the size of the launch (default), 2 = expanded
 * in LDS by 16-wave workgroups, 1 = by 4-wave workgroups, 0 = expanded into a scratch slab by a kernel of its own (rounds 1-2;
 * the slab is sized at every launch, also for streams created before the switch).
 * key "fast_tile_pitch": K2's LDS window pitch in bytes, the tile-size sweep of BASELINE.json configs[2] -- 0 = automatic
 * (48 with compile-time offsets for cells up to 36 px: the shipped shape), 48 | 64 | ... | 128 = that pitch through the
 * run-time-pitch instantiation (values below what the plan's cells need are ignored).
 * key "fast_waves_per_block": 1 (default) | 2 | 4 independent cells (waves) per K2 workgroup (measured 7 % / 15 % slower, round 5).
 * key "fast_cells_per_wave": 1 (default) ... 64 consecutive cell records a K2 wave walks, one after the other (round 5; measured
 *     slower from 2 on, profiles/r05_k2_cells_per_wave.txt: a sweep knob like the two above).
 * key "quadtree_split": K3's pass over the candidates -- 0 = inside the quadtree kernel (one launch), 1 = as a kernel of its own
 * (many small workgroups; pays for single frames and large frames), 2 = chosen per launch from the frame size and the number of
 * frames (default; the measured table is profiles/r04_k3_split_grid.txt).  PGORB_QT_SPLIT seeds it.
 * key "quadtree_threads": threads per K3 workgroup -- 0 = chosen per launch (default: 512 when the problems of a launch queue for the
 * chip or are small, 1024 when each has a CU to itself; profiles/r04_k3_threads_grid.txt), 256 | 512 | 1024 = that many.  PGORB_QT_THREADS seeds it.
 * key "fused_levels": 1 = the launch that resizes level l -> l + 1 also detects level l (csrc/fused.hip: every level read from HBM
 *     once, one wave per cell slot; bit-exact; measured SLOWER than the two launches in every form tried, profiles/r06_fused_forms.txt),
 *     0 = K1 + K2 (default).  Levels whose geometry the fused launch does not take (cells wider than 32 px, generic scale factors) run K1 + K2 either way.
 * key "pipeline_pyramid": 1 = the resize chain on a side stream beside K2, level by level (slower).
 * key "pipeline_levels": bit l set = a group of levels starts at level l; K3 / K4-6 of one group run on side streams beside K2
 * of the next (slower for every grouping measured).  0 = one launch per kernel (default).
 * key "pipeline_levels_priority": 1 = the K3 side stream is created with the highest priority (read when it is first used).
 * pgorb_get_option returns the value, or PGORB_OPTION_UNKNOWN for a null context / unknown key (-1 is a legal "match_mode"). */
#define PGORB_OPTION_UNKNOWN (-2147483647 - 1)
int  pgorb_set_option(pgorb_ctx* ctx, const char* key, int value);
int  pgorb_get_option(const pgorb_ctx* ctx, const char* key);
/* 1 when a match of `cap_per_frame` descriptors per frame takes the popcount kernels, else 0 */
int  pgorb_matcher_is_popcount(const pgorb_ctx* ctx, int cap_per_frame);

/* Host-side phases of the pgorb_extract / pgorb_extract_batch calls since the last reset, summed, in microseconds:
 * us[0] input staging + upload issue, us[1] kernel launches + download issue, us[2] wait for the GPU, us[3] results into the
 * caller's buffers.  Returns the number of calls covered; reset != 0 clears the sums (us may be NULL).  The one-frame-per-call
 * shape of the reference (Frame.cc:251-257) is measured with it: bench.py "single_frame", tools/single_frame_bench.py. */
int  pgorb_profile_host(pgorb_ctx* ctx, double* us /*[4]*/, int reset);

/* Stage taps for parity tests (host buffers, synchronous; operate on the LAST batch). */
int  pgorb_debug_level_size(const pgorb_ctx* ctx, int level, int* w, int* h);
int  pgorb_debug_level_image(pgorb_ctx* ctx, int frame, int level, uint8_t* out /* w*h */);
/* candidates of (frame, level) in device order (unordered set); x,y region-relative like
 * the reference's vToDistributeKeys (ORBextractor.cc:818-826).  Returns the count. */
int  pgorb_debug_level_candidates(pgorb_ctx* ctx, int frame, int level,
                                  int32_t* x, int32_t* y, int32_t* response, int cap);
int  pgorb_debug_level_keypoints(pgorb_ctx* ctx, int frame, int level);
/* Parity tap of the device's sin / cos contract: 64-bit checksums of pg_sincos_f over `count` consecutive
 * float bit patterns from `first_bits`, `nblocks` blocks -> out[nblocks]; compared with the oracle's for EVERY float input
 * (tests/test_gpu_parity.py::test_device_sincos_equals_the_oracle_for_every_input).  Synchronous. */
int  pgorb_debug_sincos_checksum(uint32_t first_bits, uint32_t count, int nblocks, unsigned long long* out);

#ifdef __cplusplus
}
#endif
#endif

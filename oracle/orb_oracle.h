/* oracle/orb_oracle.h -- CPU oracle for the ORB front end.  TEST INFRASTRUCTURE.
 *
 * A plain-C restatement of pilotguru's per-frame ORB path, used only as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product (libpgorb.so) never links, loads or calls anything in oracle/.
 *
 * PARITY UNPINNED for every OpenCV-backed stage: the reference's arithmetic
 * for resize / FAST / GaussianBlur / fastAtan2 / cvRound lives in un-vendored
 * OpenCV 2.4.9.1 (docker/Dockerfile:1,21), which is absent here and which the
 * reference's own tests never pin.  Those stages
 * restate the published OpenCV 2.4 algorithms.  The
 * in-tree logic (cells, quadtree, IC_Angle moments, rBRIEF, Hamming) follows
 * the reference source line by line and is cited per function in the .c file.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

typedef struct {                 /* == cv::KeyPoint (OpenCV 2.4) field order, 28 B */
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

typedef struct {                 /* one FAST candidate (region-relative coords)   */
    int32_t x, y, response;
} orc_cand;

typedef struct orc_extractor orc_extractor;

/* ORBextractor::ORBextractor, ORBextractor.cc:410-470.
 * blur_tie_mode: 0 = x86-64 OpenCV binary behaviour (SSE2 column pass rounds
 * ties to even for x < (w & ~3), scalar tail rounds half up); 1 = scalar
 * half-up everywhere. */
orc_extractor* orc_create(int nfeatures, float scale_factor, int nlevels,
                          int ini_th_fast, int min_th_fast, int blur_tie_mode);
void orc_destroy(orc_extractor*);

/* getters for the constructor tables (nlevels+1 entries where noted) */
const float* orc_scale_factors(const orc_extractor*);       /* mvScaleFactor      */
const float* orc_inv_scale_factors(const orc_extractor*);   /* mvInvScaleFactor   */
const float* orc_level_sigma2(const orc_extractor*);
const float* orc_inv_level_sigma2(const orc_extractor*);
const int*   orc_features_per_level(const orc_extractor*);  /* mnFeaturesPerLevel */
const int*   orc_umax(const orc_extractor*);                /* 16 entries         */

/* ORBextractor::operator(), ORBextractor.cc:1042-1104.
 * Returns 0 and *n keypoints (kps[cap], desc[cap*32]); <0 on error:
 *  -1 bad argument, -2 image too small for the cell grid (reference divides
 *  by zero there), -3 cap too small (nothing truncated silently). */
int orc_extract(orc_extractor*, const uint8_t* gray, int w, int h, int stride,
                orc_keypoint* kps, uint8_t* desc, int cap, int* n);

/* Stage intermediates of the LAST orc_extract call (valid until next call). */
int  orc_level_size(const orc_extractor*, int level, int* w, int* h);
const uint8_t* orc_level_image(const orc_extractor*, int level);   /* w*h, stride w */
const uint8_t* orc_level_blurred(const orc_extractor*, int level); /* NULL if level had no kp */
int  orc_level_candidates(const orc_extractor*, int level, const orc_cand** c); /* a3 order */
int  orc_level_keypoints(const orc_extractor*, int level);  /* count after quadtree */

/* Stand-alone stages (each cited in the .c file). */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride);
int  orc_ingest_geometry(const uint8_t* src, int sw, int sh, int cn, int rotate_degrees, int vertical_flip,
                         int horizontal_flip, uint8_t* dst);
void orc_fast9_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score);
int  orc_fast9_nms(const uint8_t* img, int w, int h, int stride, int threshold,
                   orc_cand* out, int cap);      /* cv::FAST(...,true); x,y window-local */
int  orc_distribute_octtree(const orc_cand* cand, int ncand, int minX, int maxX,
                            int minY, int maxY, int N, int32_t* out_idx, int cap);
float orc_fast_atan2(float y, float x);
float orc_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax);
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride,
                        uint8_t* dst, int dstride, int tie_mode);
void orc_sincos_f(float angle_rad, float* s, float* c);
void orc_orb_descriptor(const uint8_t* blurred, int stride, int x, int y,
                        float angle_deg, uint8_t desc[32]);
void orc_rgb_to_gray(const uint8_t* rgb, int w, int h, int stride, uint8_t* gray, int gstride);

/* SIMD variants of the three primitives OpenCV 2.4.9 vectorises (orb_simd.c), bit-equal to the scalar functions above */
int  orc_simd_available(void);
int  orc_set_simd(orc_extractor*, int on);
void orc_resize_linear_u8_ex(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride, int simd);
void orc_gaussian_blur7_ex(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int tie_mode, int simd);
void orc_fast9_score_map_simd(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score);
int  orc_fast9_nms_simd(const uint8_t* img, int w, int h, int stride, int threshold, orc_cand* out, int cap);
void orc_vresize_row_simd(const int* r0, const int* r1, int b0, int b1, uint8_t* out, int n);
void orc_blur_row_simd(const uint8_t* s, int w, const int K[7], int* R);
void orc_blur_col_simd(const int* const rr[7], int n, const int K[7], int tie_mode, int wvec, uint8_t* d);

/* ORBmatcher::DescriptorDistance, ORBmatcher.cc:1651-1667 */
int  orc_descriptor_distance(const uint8_t a[32], const uint8_t b[32]);
void orc_hamming_matrix(const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out);
/* best / second-best train index per query; first minimum wins (strict <). */
void orc_hamming_best2(const uint8_t* a, int na, const uint8_t* b, int nb,
                       int32_t* best_idx, uint16_t* best, uint16_t* second);

#ifdef __cplusplus
}
#endif
#endif

/* oracle/orb_oracle.c -- CPU oracle for the ORB front end.  TEST INFRASTRUCTURE.
 * See orb_oracle.h for the role and the "PARITY UNPINNED" statement.
 *
 * Every function names the reference lines it restates.  "ref:" paths are
 * relative to /root/reference/thirdparty/orb-slam2/ unless stated; "cv2.4:"
 * marks behaviour of un-vendored OpenCV 2.4.9 restated from its published
 * algorithm -- unverifiable in this container against OpenCV itself.
 * Partial third-party evidence (tests/test_oracle.py): the FAST-9 corner set and score equal
 * what scikit-image's corner_fast decides pixel for pixel, IC_Angle agrees with its corner_orientations,
 * resize / blur agree with torch / scipy float implementations to one grey level.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * Floating-point contraction MUST stay off: float -> integer roundings below
 * decide which pixel a BRIEF tap reads.
 */
#include "orb_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const int8_t k_pattern[256 * 4] = {
#include "orb_pattern31.inc"
};

enum { PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19 }; /* ref: src/ORBextractor.cc:72-74 */

/* cv2.4: cvRound = SSE2 cvtsd2si = round half to even; cvFloor / cvCeil exact. */
static int cv_round(double v) { return (int)lrint(v); }
static int cv_floor(double v) { int i = (int)v; return i - (v < i); }
static int cv_ceil(double v)  { int i = (int)v; return i + (v > i); }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------ */
struct orc_extractor {
    int nfeatures, nlevels, iniThFAST, minThFAST, blur_tie_mode;
    int simd;                                 /* orc_set_simd: 1 = the three primitives OpenCV 2.4.9 vectorises run their SIMD variants (orb_simd.c) */
    double scaleFactor;                       /* ref: include/ORBextractor.h:95 (double member) */
    float mvScaleFactor[ORC_MAX_LEVELS + 1], mvInvScaleFactor[ORC_MAX_LEVELS + 1];
    float mvLevelSigma2[ORC_MAX_LEVELS + 1], mvInvLevelSigma2[ORC_MAX_LEVELS + 1];
    int mnFeaturesPerLevel[ORC_MAX_LEVELS + 1];
    int umax[HALF_PATCH_SIZE + 1];
    /* per-call intermediates */
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    uint8_t* img[ORC_MAX_LEVELS];
    uint8_t* blur[ORC_MAX_LEVELS];
    orc_cand* cand[ORC_MAX_LEVELS];
    int ncand[ORC_MAX_LEVELS];
    int nkp[ORC_MAX_LEVELS];
};

/* ref: src/ORBextractor.cc:410-470 (constructor) */
orc_extractor* orc_create(int nfeatures, float scale_factor, int nlevels,
                          int ini_th_fast, int min_th_fast, int blur_tie_mode)
{
    if (nlevels < 1 || nlevels > ORC_MAX_LEVELS || nfeatures < 1) return NULL;
    orc_extractor* e = (orc_extractor*)calloc(1, sizeof(*e));
    e->nfeatures = nfeatures; e->nlevels = nlevels;
    e->iniThFAST = ini_th_fast; e->minThFAST = min_th_fast;
    e->blur_tie_mode = blur_tie_mode;
    e->scaleFactor = scale_factor;                       /* float -> double member */
    e->mvScaleFactor[0] = 1.0f; e->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i <= nlevels; i++) {                 /* :419-423, float*double -> float */
        e->mvScaleFactor[i] = (float)(e->mvScaleFactor[i - 1] * e->scaleFactor);
        e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];
    }
    for (int i = 0; i <= nlevels; i++) {                 /* :427-431 */
        e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
        e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
    }
    float factor = (float)(1.0f / e->scaleFactor);       /* :436 */
    float nDesired = nfeatures * (1 - factor) /
                     (1 - (float)pow((double)factor, (double)nlevels));   /* :437 */
    int sum = 0;
    for (int level = 0; level < nlevels; level++) {      /* :440-445 */
        e->mnFeaturesPerLevel[level] = cv_round(nDesired);
        sum += e->mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    e->mnFeaturesPerLevel[nlevels] = imax(nfeatures - sum, 0);   /* :446, unused slot */

    /* :454-469 circular patch row extents */
    int v, v0;
    int vmax = cv_floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}

static void free_intermediates(orc_extractor* e)
{
    for (int l = 0; l < ORC_MAX_LEVELS; l++) {
        free(e->img[l]); free(e->blur[l]); free(e->cand[l]);
        e->img[l] = e->blur[l] = NULL; e->cand[l] = NULL;
        e->ncand[l] = e->nkp[l] = 0;
    }
}
void orc_destroy(orc_extractor* e) { if (e) { free_intermediates(e); free(e); } }
/* 1 = FAST, the vertical resize pass and both blur passes through their SIMD variants (orb_simd.c; the same results bit for bit):
 * what bench.py's cpu_baseline reports as "port+simd".  Returns what was set: 0 when this CPU cannot run them. */
int orc_set_simd(orc_extractor* e, int on) { e->simd = (on && orc_simd_available()) ? 1 : 0; return e->simd; }

const float* orc_scale_factors(const orc_extractor* e) { return e->mvScaleFactor; }
const float* orc_inv_scale_factors(const orc_extractor* e) { return e->mvInvScaleFactor; }
const float* orc_level_sigma2(const orc_extractor* e) { return e->mvLevelSigma2; }
const float* orc_inv_level_sigma2(const orc_extractor* e) { return e->mvInvLevelSigma2; }
const int* orc_features_per_level(const orc_extractor* e) { return e->mnFeaturesPerLevel; }
const int* orc_umax(const orc_extractor* e) { return e->umax; }
int orc_level_size(const orc_extractor* e, int l, int* w, int* h)
{ if (l < 0 || l >= e->nlevels) return -1; *w = e->lw[l]; *h = e->lh[l]; return 0; }
const uint8_t* orc_level_image(const orc_extractor* e, int l) { return e->img[l]; }
const uint8_t* orc_level_blurred(const orc_extractor* e, int l) { return e->blur[l]; }
int orc_level_candidates(const orc_extractor* e, int l, const orc_cand** c) { *c = e->cand[l]; return e->ncand[l]; }
int orc_level_keypoints(const orc_extractor* e, int l) { return e->nkp[l]; }

/* ------------------------------------------------------------------------ */
/* cv2.4: cv::resize(..., INTER_LINEAR) for CV_8UC1 (imgproc/imgwarp.cpp:
 * resize() coefficient set-up, HResizeLinear<uchar,int,short,2048>,
 * VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>).  Called at
 * ref: src/ORBextractor.cc:1119.  Source is the level ROI only. */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride)
{
    orc_resize_linear_u8_ex(src, sw, sh, sstride, dst, dw, dh, dstride, 0);
}
/* simd != 0: the vertical pass through orb_simd.c (cv2.4: VResizeLinearVec_32s8u); same result bit for bit */
void orc_resize_linear_u8_ex(const uint8_t* src, int sw, int sh, int sstride,
                             uint8_t* dst, int dw, int dh, int dstride, int simd)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = imin(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        int a0 = cv_round(c0 * 2048), a1 = cv_round(c1 * 2048);   /* saturate_cast<short> */
        ialpha[dx * 2] = (short)imax(-32768, imin(32767, a0));
        ialpha[dx * 2 + 1] = (short)imax(-32768, imin(32767, a1));
    }
    int* rows[2];
    rows[0] = (int*)malloc(sizeof(int) * dw);
    rows[1] = (int*)malloc(sizeof(int) * dw);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        float c0 = 1.f - fy, c1 = fy;
        short b0 = (short)imax(-32768, imin(32767, cv_round(c0 * 2048)));
        short b1 = (short)imax(-32768, imin(32767, cv_round(c1 * 2048)));
        for (int k = 0; k < 2; k++) {                       /* clip(sy+k, 0, sh) */
            int r = sy + k; if (r < 0) r = 0; if (r >= sh) r = sh - 1;
            const uint8_t* S = src + (size_t)r * sstride;
            int* D = rows[k];
            int dx = 0;
            for (; dx < xmax; dx++) {
                int sx = xofs[dx];
                D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
            }
            for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
        }
        uint8_t* out = dst + (size_t)dy * dstride;
        if (simd) { orc_vresize_row_simd(rows[0], rows[1], b0, b1, out, dw); continue; }
        for (int x = 0; x < dw; x++)
            out[x] = (uint8_t)((((b0 * (rows[0][x] >> 4)) >> 16) +
                                ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2);
    }
    free(rows[0]); free(rows[1]); free(xofs); free(ialpha);
}

/* ------------------------------------------------------------------------ */
/* cv2.4: cv::FAST(img, kps, threshold, nonmaxSuppression=true), 16-pixel
 * ring, 9 contiguous (features2d/fast.cpp FAST_t<16>, fast_score.cpp
 * cornerScore<16>).  Called per cell at ref: src/ORBextractor.cc:809-815.
 * Output order is row-major; x,y are window-local. */
static const int k_ring[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int fast_is_corner(const int d[25], int t)
{
    /* d[k] = v - ring[k]; "darker" run: d > t ; "brighter" run: d < -t */
    int run_d = 0, run_b = 0;
    for (int k = 0; k < 25; k++) {
        run_d = (d[k] > t) ? run_d + 1 : 0;
        run_b = (d[k] < -t) ? run_b + 1 : 0;
        if (run_d >= 9 || run_b >= 9) return 1;
    }
    return 0;
}
static int fast_corner_score(const int d[25], int threshold)
{
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = imin(d[k + 1], d[k + 2]);
        for (int m = 3; m <= 8; m++) a = imin(a, d[k + m]);
        a0 = imax(a0, imin(a, d[k]));
        a0 = imax(a0, imin(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = imax(d[k + 1], d[k + 2]);
        for (int m = 3; m <= 8; m++) b = imax(b, d[k + m]);
        b0 = imin(b0, imax(b, d[k]));
        b0 = imin(b0, imax(b, d[k + 9]));
    }
    return -b0 - 1;
}
/* corner scores of every pixel (0 = not a FAST-9 corner at `threshold`), before NMS; score is w*h,
 * zeroed here.  (Also the hook for the cross-check of the corner SET against scikit-image's
 * independent corner_fast, tests/golden/make_skimage_fast9.py.) */
void orc_fast9_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score)
{
    memset(score, 0, (size_t)w * h);
    if (w < 7 || h < 7) return;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            int v = p[0], d[25];
            {   /* any 9-arc holds a pixel of every opposite pair: cheap reject (same result) */
                const int lo = v - threshold, hi = v + threshold;
                const int r0 = p[3 * stride], r8 = p[-3 * stride], r4 = p[3], r12 = p[-3];
                if (!(((r0 < lo || r8 < lo) && (r4 < lo || r12 < lo)) ||
                      ((r0 > hi || r8 > hi) && (r4 > hi || r12 > hi)))) continue;
            }
            for (int k = 0; k < 25; k++)
                d[k] = v - p[k_ring[k & 15][1] * stride + k_ring[k & 15][0]];
            if (fast_is_corner(d, threshold))
                score[(size_t)y * w + x] = (uint8_t)fast_corner_score(d, threshold);
        }
}
int orc_fast9_nms(const uint8_t* img, int w, int h, int stride, int threshold,
                  orc_cand* out, int cap)
{
    if (w < 7 || h < 7) return 0;
    uint8_t* score = (uint8_t*)malloc((size_t)w * h);
    orc_fast9_score_map(img, w, h, stride, threshold, score);
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = score[(size_t)y * w + x];
            if (!s) continue;   /* non-corner (a corner's score is >= threshold >= 1 here) */
            const uint8_t* r0 = score + (size_t)(y - 1) * w + x;
            const uint8_t* r1 = score + (size_t)y * w + x;
            const uint8_t* r2 = score + (size_t)(y + 1) * w + x;
            if (s > r1[1] && s > r1[-1] && s > r0[-1] && s > r0[0] && s > r0[1] &&
                s > r2[-1] && s > r2[0] && s > r2[1]) {
                if (n < cap) { out[n].x = x; out[n].y = y; out[n].response = s; }
                n++;
            }
        }
    free(score);
    return n;
}

/* ------------------------------------------------------------------------ */
/* ref: src/ORBextractor.cc:481-537 ExtractorNode::DivideNode and :539-763
 * ORBextractor::DistributeOctTree.  std::list is restated as an index-linked
 * list over an append-only node pool, so a node's pool index IS its creation
 * sequence number.  PARITY CONTRACT for the allocator-dependent tie at :684
 * (sort of pair<int,ExtractorNode*> compares pointer values): equal-sized
 * nodes are ordered by creation sequence, later created = larger = split
 * first (what a never-reusing bump allocator gives; SURVEY.md hard part 2). */
typedef struct {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int kbeg, kcnt;            /* slice of the key pool (indices into cand[]) */
    int prev, next;            /* list links, -1 = none                         */
    int bNoMore;
} qt_node;
typedef struct {
    qt_node* nodes; int nn, ncap;
    int* keys; int nk, kcap;
    int head, tail, size;
    const orc_cand* cand;
} qt_state;

static int qt_new_node(qt_state* s)
{
    if (s->nn == s->ncap) { s->ncap *= 2; s->nodes = (qt_node*)realloc(s->nodes, sizeof(qt_node) * s->ncap); }
    memset(&s->nodes[s->nn], 0, sizeof(qt_node));
    s->nodes[s->nn].prev = s->nodes[s->nn].next = -1;
    return s->nn++;
}
static void qt_reserve_keys(qt_state* s, int extra)
{
    if (s->nk + extra > s->kcap) {
        while (s->nk + extra > s->kcap) s->kcap *= 2;
        s->keys = (int*)realloc(s->keys, sizeof(int) * s->kcap);
    }
}
static void qt_push_front(qt_state* s, int n)
{
    s->nodes[n].prev = -1; s->nodes[n].next = s->head;
    if (s->head >= 0) s->nodes[s->head].prev = n; else s->tail = n;
    s->head = n; s->size++;
}
static void qt_push_back(qt_state* s, int n)
{
    s->nodes[n].next = -1; s->nodes[n].prev = s->tail;
    if (s->tail >= 0) s->nodes[s->tail].next = n; else s->head = n;
    s->tail = n; s->size++;
}
static int qt_erase(qt_state* s, int n)          /* returns the following node */
{
    int p = s->nodes[n].prev, nx = s->nodes[n].next;
    if (p >= 0) s->nodes[p].next = nx; else s->head = nx;
    if (nx >= 0) s->nodes[nx].prev = p; else s->tail = p;
    s->size--;
    return nx;
}
/* DivideNode (:481-537): creates the four children in the pool (n1..n4 get
 * consecutive indices c, c+1, c+2, c+3) WITHOUT linking them. */
static int qt_divide(qt_state* s, int parent)
{
    int c = qt_new_node(s); qt_new_node(s); qt_new_node(s); qt_new_node(s);
    qt_node P = s->nodes[parent];
    const int halfX = (int)ceilf((float)(P.URx - P.ULx) / 2);
    const int halfY = (int)ceilf((float)(P.BRy - P.ULy) / 2);
    qt_node* n1 = &s->nodes[c]; qt_node* n2 = n1 + 1; qt_node* n3 = n1 + 2; qt_node* n4 = n1 + 3;
    n1->ULx = P.ULx; n1->ULy = P.ULy; n1->URx = P.ULx + halfX; n1->URy = P.ULy;
    n1->BLx = P.ULx; n1->BLy = P.ULy + halfY; n1->BRx = P.ULx + halfX; n1->BRy = P.ULy + halfY;
    n2->ULx = n1->URx; n2->ULy = n1->URy; n2->URx = P.URx; n2->URy = P.URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy; n2->BRx = P.URx; n2->BRy = P.ULy + halfY;
    n3->ULx = n1->BLx; n3->ULy = n1->BLy; n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = P.BLx; n3->BLy = P.BLy; n3->BRx = n1->BRx; n3->BRy = P.BLy;
    n4->ULx = n3->URx; n4->ULy = n3->URy; n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy; n4->BRx = P.BRx; n4->BRy = P.BRy;
    /* stable 4-way partition of the parent's keys */
    int cnt[4] = {0, 0, 0, 0};
    int* q = (int*)malloc(sizeof(int) * (P.kcnt > 0 ? P.kcnt : 1));
    for (int i = 0; i < P.kcnt; i++) {
        const orc_cand* kp = &s->cand[s->keys[P.kbeg + i]];
        int which;
        if ((float)kp->x < (float)n1->URx) which = ((float)kp->y < (float)n1->BRy) ? 0 : 2;
        else which = ((float)kp->y < (float)n1->BRy) ? 1 : 3;
        q[i] = which; cnt[which]++;
    }
    qt_reserve_keys(s, P.kcnt);
    n1 = &s->nodes[c];                                   /* (pool may have moved) */
    int beg[4]; beg[0] = s->nk; beg[1] = beg[0] + cnt[0]; beg[2] = beg[1] + cnt[1]; beg[3] = beg[2] + cnt[2];
    int pos[4] = {beg[0], beg[1], beg[2], beg[3]};
    for (int i = 0; i < P.kcnt; i++) s->keys[pos[q[i]]++] = s->keys[P.kbeg + i];
    s->nk += P.kcnt;
    for (int k = 0; k < 4; k++) {
        n1[k].kbeg = beg[k]; n1[k].kcnt = cnt[k];
        if (cnt[k] == 1) n1[k].bNoMore = 1;
    }
    free(q);
    return c;
}
typedef struct { int size, node; } qt_pair;
static int qt_pair_cmp(const void* a, const void* b)
{
    const qt_pair* x = (const qt_pair*)a; const qt_pair* y = (const qt_pair*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->node < y->node ? -1 : (x->node > y->node);
}
/* Adds the non-empty children of `parent` to the list front in n1..n4 order
 * and records the expandable ones (:621-660 / :691-726). */
static void qt_split_and_link(qt_state* s, int parent, qt_pair* vec, int* nvec, int* nToExpand)
{
    int c = qt_divide(s, parent);
    for (int k = 0; k < 4; k++) {
        if (s->nodes[c + k].kcnt > 0) {
            qt_push_front(s, c + k);
            if (s->nodes[c + k].kcnt > 1) {
                if (nToExpand) (*nToExpand)++;
                vec[*nvec].size = s->nodes[c + k].kcnt; vec[*nvec].node = c + k; (*nvec)++;
            }
        }
    }
}
/* Returns the number of selected keys; out_idx[i] = index into cand[] of the
 * i-th result in the reference's output order. */
int orc_distribute_octtree(const orc_cand* cand, int ncand, int minX, int maxX,
                           int minY, int maxY, int N, int32_t* out_idx, int cap)
{
    if (ncand <= 0) return 0;
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));   /* :543 */
    if (nIni < 1) return -2;                    /* reference indexes an empty vector here */
    const float hX = (float)(maxX - minX) / nIni;                        /* :545 */
    qt_state s; memset(&s, 0, sizeof(s));
    s.ncap = 64 + 8 * (N > 0 ? N : 1); s.nodes = (qt_node*)malloc(sizeof(qt_node) * s.ncap);
    s.kcap = 4 * ncand + 16; s.keys = (int*)malloc(sizeof(int) * s.kcap);
    s.head = s.tail = -1; s.cand = cand;

    /* :552-570 initial nodes + assignment (two passes keep each root's keys contiguous) */
    int* root_of = (int*)malloc(sizeof(int) * ncand);
    int* rcnt = (int*)calloc(nIni, sizeof(int));
    for (int i = 0; i < ncand; i++) {
        int r = (int)((float)cand[i].x / hX);
        if (r < 0 || r >= nIni) { free(root_of); free(rcnt); free(s.nodes); free(s.keys); return -1; }
        root_of[i] = r; rcnt[r]++;
    }
    for (int i = 0; i < nIni; i++) {
        int n = qt_new_node(&s);
        qt_node* ni = &s.nodes[n];
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        ni->kbeg = s.nk; ni->kcnt = 0; s.nk += rcnt[i];
        qt_push_back(&s, n);
    }
    for (int i = 0; i < ncand; i++) {
        qt_node* ni = &s.nodes[root_of[i]];
        s.keys[ni->kbeg + ni->kcnt++] = i;
    }
    free(root_of); free(rcnt);
    /* :572-585 */
    for (int lit = s.head; lit >= 0;) {
        if (s.nodes[lit].kcnt == 1) { s.nodes[lit].bNoMore = 1; lit = s.nodes[lit].next; }
        else if (s.nodes[lit].kcnt == 0) lit = qt_erase(&s, lit);
        else lit = s.nodes[lit].next;
    }
    qt_pair* vec = (qt_pair*)malloc(sizeof(qt_pair) * (4 * (size_t)ncand + 16));
    qt_pair* prevvec = (qt_pair*)malloc(sizeof(qt_pair) * (4 * (size_t)ncand + 16));
    int nvec = 0, bFinish = 0;
    while (!bFinish) {                                                /* :594-739 */
        int prevSize = s.size, nToExpand = 0;
        nvec = 0;
        for (int lit = s.head; lit >= 0;) {
            if (s.nodes[lit].bNoMore) { lit = s.nodes[lit].next; continue; }
            qt_split_and_link(&s, lit, vec, &nvec, &nToExpand);
            lit = qt_erase(&s, lit);
        }
        if (s.size >= N || s.size == prevSize) bFinish = 1;           /* :669 */
        else if (s.size + nToExpand * 3 > N) {                        /* :673 */
            while (!bFinish) {
                prevSize = s.size;
                int nprev = nvec; memcpy(prevvec, vec, sizeof(qt_pair) * nvec);
                nvec = 0;
                qsort(prevvec, nprev, sizeof(qt_pair), qt_pair_cmp);  /* :684 (see contract) */
                for (int j = nprev - 1; j >= 0; j--) {
                    qt_split_and_link(&s, prevvec[j].node, vec, &nvec, NULL);
                    qt_erase(&s, prevvec[j].node);
                    if (s.size >= N) break;                           /* :730 */
                }
                if (s.size >= N || s.size == prevSize) bFinish = 1;   /* :734 */
            }
        }
    }
    /* :741-760 best response per node, first maximum wins */
    int nout = 0;
    for (int lit = s.head; lit >= 0; lit = s.nodes[lit].next) {
        const qt_node* nd = &s.nodes[lit];
        int best = s.keys[nd->kbeg];
        float maxResponse = (float)cand[best].response;
        for (int k = 1; k < nd->kcnt; k++) {
            int ki = s.keys[nd->kbeg + k];
            if ((float)cand[ki].response > maxResponse) { best = ki; maxResponse = (float)cand[ki].response; }
        }
        if (nout < cap) out_idx[nout] = best;
        nout++;
    }
    free(vec); free(prevvec); free(s.nodes); free(s.keys);
    return nout;
}

/* ------------------------------------------------------------------------ */
/* cv2.4: cv::fastAtan2 (core/mathfuncs.cpp), degrees in [0,360]. */
float orc_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ref: src/ORBextractor.cc:77-104 IC_Angle */
float orc_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------------ */
/* cv2.4: cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) on
 * CV_8UC1 (imgproc/smooth.cpp createGaussianFilter -> getGaussianKernel
 * CV_32F; filter.cpp createSeparableLinearFilter fixed-point bits=8 per
 * pass; RowFilter<uchar,int>, SymmColumnFilter<FixedPtCastEx<int,uchar>>
 * with SymmColumnVec_32s8u).  Called at ref: src/ORBextractor.cc:1085. */
static void gauss7_kernel_q8(int K[7])
{
    float cf[7]; double sum = 0;
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        double t = exp(-0.5 / (2.0 * 2.0) * x * x);
        cf[i] = (float)t; sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        K[i] = cv_round((double)(cf[i] * 256.f));     /* Mat::convertTo(CV_32S, 256) */
    }
}
static int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride,
                        uint8_t* dst, int dstride, int tie_mode)
{
    orc_gaussian_blur7_ex(src, w, h, sstride, dst, dstride, tie_mode, 0);
}
/* simd != 0: both passes through orb_simd.c (cv2.4: RowVec_8u32s, SymmColumnVec_32s8u); same result bit for bit */
void orc_gaussian_blur7_ex(const uint8_t* src, int w, int h, int sstride,
                           uint8_t* dst, int dstride, int tie_mode, int simd)
{
    int K[7]; gauss7_kernel_q8(K);
    int* R = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        if (simd && w >= 7) {
            orc_blur_row_simd(s, w, K, R + (size_t)y * w);            /* the interior; the six border columns below */
            for (int x = 0; x < w; x = (x == 2 ? w - 3 : x + 1)) {
                int acc = 0;
                for (int i = 0; i < 7; i++) acc += K[i] * s[reflect101(x + i - 3, w)];
                R[(size_t)y * w + x] = acc;
            }
            continue;
        }
        for (int x = 0; x < w; x++) {
            int acc = 0;
            if (x >= 3 && x + 3 < w)
                acc = K[0] * (s[x - 3] + s[x + 3]) + K[1] * (s[x - 2] + s[x + 2]) + K[2] * (s[x - 1] + s[x + 1]) + K[3] * s[x];
            else
                for (int i = 0; i < 7; i++) acc += K[i] * s[reflect101(x + i - 3, w)];
            R[(size_t)y * w + x] = acc;
        }
    }
    const int wvec = w & ~3;          /* SSE2 column pass covers x < wvec */
    for (int y = 0; y < h; y++) {
        uint8_t* d = dst + (size_t)y * dstride;
        const int* rr[7];
        for (int i = 0; i < 7; i++) rr[i] = R + (size_t)reflect101(y + i - 3, h) * w;
        if (simd) { orc_blur_col_simd(rr, w, K, tie_mode, wvec, d); continue; }
        for (int x = 0; x < w; x++) {
            int C = K[0] * (rr[0][x] + rr[6][x]) + K[1] * (rr[1][x] + rr[5][x]) + K[2] * (rr[2][x] + rr[4][x]) + K[3] * rr[3][x];
            int v = (C + 32768) >> 16;                 /* FixedPtCastEx: half up */
            if (tie_mode == 0 && x < wvec && (C & 0xFFFF) == 0x8000)
                v &= ~1;                                /* cvtps2dq: tie -> even  */
            d[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    free(R);
}

/* ------------------------------------------------------------------------ */
/* sin/cos of the keypoint angle.  ref: src/ORBextractor.cc:112-113 calls
 * cos(float)/sin(float) = the platform libm's cosf/sinf, whose last-bit
 * behaviour differs between glibc versions.  PARITY CONTRACT: a and b are
 * sin/cos evaluated in IEEE double by the fixed sequence below (no FMA) and
 * rounded once to float -- reproducible bit-for-bit on host and GPU, and
 * equal to the correctly rounded value except for ~1e-9 of inputs.  The
 * measured disagreement with this container's glibc cosf/sinf is recorded in
 * DESIGN.md section 5 (tools/sincos_sweep.c, profiles/r02_sincos_sweep.txt). */
void orc_sincos_f(float angle_rad, float* s_out, float* c_out)
{
    const double INV_PIO2 = 0.63661977236758138243;
    const double PIO2_HI = 1.57079632673412561417e+00;   /* first 33 bits of pi/2 */
    const double PIO2_LO = 6.07710050650619224932e-11;   /* pi/2 - PIO2_HI        */
    double x = (double)angle_rad;
    double kd = floor(x * INV_PIO2 + 0.5);
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    double r2 = r * r;
    double ps = -1.0 / 355687428096000.0;               /* -1/17! */
    ps = ps * r2 + 1.0 / 1307674368000.0;               /*  1/15! */
    ps = ps * r2 - 1.0 / 6227020800.0;                  /* -1/13! */
    ps = ps * r2 + 1.0 / 39916800.0;                    /*  1/11! */
    ps = ps * r2 - 1.0 / 362880.0;                      /* -1/9!  */
    ps = ps * r2 + 1.0 / 5040.0;                        /*  1/7!  */
    ps = ps * r2 - 1.0 / 120.0;                         /* -1/5!  */
    ps = ps * r2 + 1.0 / 6.0;                           /*  1/3!  */
    double sn = r - (r * r2) * ps;
    double pc = -1.0 / 6402373705728000.0;              /* -1/18! */
    pc = pc * r2 + 1.0 / 20922789888000.0;              /*  1/16! */
    pc = pc * r2 - 1.0 / 87178291200.0;                 /* -1/14! */
    pc = pc * r2 + 1.0 / 479001600.0;                   /*  1/12! */
    pc = pc * r2 - 1.0 / 3628800.0;                     /* -1/10! */
    pc = pc * r2 + 1.0 / 40320.0;                       /*  1/8!  */
    pc = pc * r2 - 1.0 / 720.0;                         /* -1/6!  */
    pc = pc * r2 + 1.0 / 24.0;                          /*  1/4!  */
    pc = pc * r2 - 0.5;
    double cs = 1.0 + r2 * pc;
    long long k = (long long)kd;
    double s, c;
    switch ((int)(k & 3)) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s; *c_out = (float)c;
}

/* checksum of orc_sincos_f over `count` consecutive float bit patterns (the GPU's k_sincos_checksum folds
 * pg_sincos_f the same way): the exhaustive device-vs-host check of the sin/cos contract */
uint64_t orc_sincos_checksum(uint32_t first_bits, uint32_t count)
{
    uint64_t acc = 0;
    for (uint32_t i = 0; i < count; i++) {
        union { uint32_t u; float f; } in, so, co;
        in.u = first_bits + i;
        orc_sincos_f(in.f, &so.f, &co.f);
        acc += ((uint64_t)so.u * 0x9E3779B1ull + co.u) * (2ull * i + 1ull);
    }
    return acc;
}

/* ref: src/ORBextractor.cc:107-147 computeOrbDescriptor (unfused float math) */
void orc_orb_descriptor(const uint8_t* img, int stride, int x, int y,
                        float angle_deg, uint8_t desc[32])
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    float a, b;
    orc_sincos_f(angle, &b, &a);
    const uint8_t* center = img + (size_t)y * stride + x;
    const int8_t* pat = k_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int m = 0; m < 8; m++) {
            int px0 = pat[4 * m], py0 = pat[4 * m + 1], px1 = pat[4 * m + 2], py1 = pat[4 * m + 3];
            float r0 = px0 * b; float r0b = py0 * a; float c0 = px0 * a; float c0b = py0 * b;
            float r1 = px1 * b; float r1b = py1 * a; float c1 = px1 * a; float c1b = py1 * b;
            int t0 = center[cv_round((double)(r0 + r0b)) * stride + cv_round((double)(c0 - c0b))];
            int t1 = center[cv_round((double)(r1 + r1b)) * stride + cv_round((double)(c1 - c1b))];
            val |= (t0 < t1) << m;
        }
        desc[i] = (uint8_t)val;
    }
}

/* cv2.4: cvtColor RGB2GRAY 8U (ref caller: src/Tracking.cc:247-260) */
void orc_rgb_to_gray(const uint8_t* rgb, int w, int h, int stride, uint8_t* gray, int gstride)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = rgb + (size_t)y * stride + 3 * x;
            gray[(size_t)y * gstride + x] = (uint8_t)((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14);
        }
}

/* ref (pilotguru): src/io/image_sequence_reader.cc:186-205 (rotation from the video metadata, as
 * cv::transpose + cv::flip), :53-58 + :212-222 (optional flips, cv::flip axis 0 = around the x axis
 * = rows reversed, 1 = columns reversed, -1 = both).  Restated literally as a sequence of whole-image
 * steps; `cn` interleaved channels per pixel are carried along.  dst is dw x dh with
 * (dw, dh) = (sh, sw) for 90 / 270.  Returns 0, or -1 for an unsupported angle (:203-207). */
static void img_transpose(const uint8_t* a, int w, int h, int cn, uint8_t* t)        /* t is h x w -> w rows of h */
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            memcpy(t + ((size_t)x * h + y) * cn, a + ((size_t)y * w + x) * cn, (size_t)cn);
}
static void img_flip(const uint8_t* a, int w, int h, int cn, int axis, uint8_t* f)  /* cv::flip */
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int sy = (axis == 0 || axis == -1) ? h - 1 - y : y;
            const int sx = (axis == 1 || axis == -1) ? w - 1 - x : x;
            memcpy(f + ((size_t)y * w + x) * cn, a + ((size_t)sy * w + sx) * cn, (size_t)cn);
        }
}
int orc_ingest_geometry(const uint8_t* src, int sw, int sh, int cn, int rotate_degrees, int vertical_flip,
                        int horizontal_flip, uint8_t* dst)
{
    const size_t bytes = (size_t)sw * sh * cn;
    uint8_t* t = (uint8_t*)malloc(bytes);
    uint8_t* r = (uint8_t*)malloc(bytes);
    int w = sw, h = sh;
    switch (rotate_degrees) {
    case 0: memcpy(r, src, bytes); break;
    case 90: img_transpose(src, sw, sh, cn, t); w = sh; h = sw; img_flip(t, w, h, cn, 0, r); break;
    case 180: img_flip(src, sw, sh, cn, -1, r); break;
    case 270: img_transpose(src, sw, sh, cn, t); w = sh; h = sw; img_flip(t, w, h, cn, 1, r); break;
    default: free(t); free(r); return -1;
    }
    if (vertical_flip || horizontal_flip) {
        const int axis = (vertical_flip && horizontal_flip) ? -1 : (vertical_flip ? 0 : 1);       /* MakeFlipAxis */
        img_flip(r, w, h, cn, axis, dst);
    } else {
        memcpy(dst, r, bytes);
    }
    free(t); free(r);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* ref: src/ORBextractor.cc:765-852 ComputeKeyPointsOctTree, cell loop part */
static int detect_level_cells(const orc_extractor* e, const uint8_t* img, int cols, int rows,
                              orc_cand** out_cand)
{
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = cols - EDGE_THRESHOLD + 3, maxBorderY = rows - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return -2;
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
    int cap = 4096, n = 0;
    orc_cand* all = (orc_cand*)malloc(sizeof(orc_cand) * cap);
    int ccap = (wCell + 6) * (hCell + 6);
    orc_cand* cell = (orc_cand*)malloc(sizeof(orc_cand) * ccap);
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
            const uint8_t* win = img + (size_t)y0 * cols + x0;
            int (*const nms)(const uint8_t*, int, int, int, int, orc_cand*, int) = e->simd ? orc_fast9_nms_simd : orc_fast9_nms;
            int nc = nms(win, x1 - x0, y1 - y0, cols, e->iniThFAST, cell, ccap);
            if (nc == 0) nc = nms(win, x1 - x0, y1 - y0, cols, e->minThFAST, cell, ccap);
            for (int k = 0; k < nc; k++) {
                if (n == cap) { cap *= 2; all = (orc_cand*)realloc(all, sizeof(orc_cand) * cap); }
                all[n].x = cell[k].x + j * wCell;
                all[n].y = cell[k].y + i * hCell;
                all[n].response = cell[k].response;
                n++;
            }
        }
    }
    free(cell);
    *out_cand = all;
    return n;
}

/* ref: src/ORBextractor.cc:1042-1104 operator(), :1106-1131 ComputePyramid,
 * :765-852 ComputeKeyPointsOctTree.  The 19-px reflect border ComputePyramid
 * adds is never read on this path, so levels are kept
 * unpadded. */
int orc_extract(orc_extractor* e, const uint8_t* gray, int w, int h, int stride,
                orc_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (!e || !n_out) return -1;
    *n_out = 0;
    if (!gray || w <= 0 || h <= 0) return 0;            /* :1045 empty image: silent return */
    free_intermediates(e);
    const int L = e->nlevels;
    for (int level = 0; level < L; ++level) {           /* :1108-1129 */
        float scale = e->mvInvScaleFactor[level];
        e->lw[level] = cv_round((double)((float)w * scale));
        e->lh[level] = cv_round((double)((float)h * scale));
        if (e->lw[level] - 2 * EDGE_THRESHOLD + 6 < 30 || e->lh[level] - 2 * EDGE_THRESHOLD + 6 < 30) return -2;
    }
    for (int level = 0; level < L; ++level) {
        e->img[level] = (uint8_t*)malloc((size_t)e->lw[level] * e->lh[level]);
        if (level == 0)
            for (int y = 0; y < h; y++) memcpy(e->img[0] + (size_t)y * w, gray + (size_t)y * stride, w);
        else
            orc_resize_linear_u8_ex(e->img[level - 1], e->lw[level - 1], e->lh[level - 1], e->lw[level - 1],
                                    e->img[level], e->lw[level], e->lh[level], e->lw[level], e->simd);
    }
    int total = 0;
    orc_keypoint* lk[ORC_MAX_LEVELS] = {0};
    for (int level = 0; level < L; ++level) {           /* :771-847 */
        const int cols = e->lw[level], rows = e->lh[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = cols - EDGE_THRESHOLD + 3, maxBorderY = rows - EDGE_THRESHOLD + 3;
        int nc = detect_level_cells(e, e->img[level], cols, rows, &e->cand[level]);
        if (nc < 0) return nc;
        e->ncand[level] = nc;
        const int N = e->mnFeaturesPerLevel[level];
        int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc + 4));
        int nk = orc_distribute_octtree(e->cand[level], nc, minBorderX, maxBorderX,
                                        minBorderY, maxBorderY, N, idx, nc + 4);
        if (nk < 0) { free(idx); return nk; }
        e->nkp[level] = nk;
        const int scaledPatchSize = (int)(PATCH_SIZE * e->mvScaleFactor[level]);     /* :836 */
        lk[level] = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)(nk + 1));
        for (int i = 0; i < nk; i++) {
            const orc_cand* c = &e->cand[level][idx[i]];
            orc_keypoint* k = &lk[level][i];
            k->x = (float)c->x + minBorderX; k->y = (float)c->y + minBorderY;       /* :842-843 */
            k->octave = level; k->size = (float)scaledPatchSize;
            k->response = (float)c->response; k->class_id = -1;
            k->angle = orc_ic_angle(e->img[level], cols, cv_round(k->x), cv_round(k->y), e->umax); /* :850-851 */
        }
        free(idx);
        total += nk;
    }
    if (total > cap) { for (int l = 0; l < L; l++) free(lk[l]); return -3; }
    int offset = 0;
    for (int level = 0; level < L; ++level) {           /* :1075-1103 */
        const int nk = e->nkp[level];
        if (nk == 0) { free(lk[level]); continue; }
        const int cols = e->lw[level], rows = e->lh[level];
        e->blur[level] = (uint8_t*)malloc((size_t)cols * rows);
        orc_gaussian_blur7_ex(e->img[level], cols, rows, cols, e->blur[level], cols, e->blur_tie_mode, e->simd);
        for (int i = 0; i < nk; i++) {
            orc_keypoint* k = &lk[level][i];
            orc_orb_descriptor(e->blur[level], cols, cv_round(k->x), cv_round(k->y), k->angle,
                               desc + (size_t)(offset + i) * 32);
            if (level != 0) {                           /* :1094-1100 pt *= scale */
                float scale = e->mvScaleFactor[level];
                k->x = k->x * scale; k->y = k->y * scale;
            }
            kps[offset + i] = *k;
        }
        offset += nk;
        free(lk[level]);
    }
    *n_out = total;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* ref: src/ORBmatcher.cc:1651-1667 DescriptorDistance (SWAR popcount over 8
 * 32-bit words); identical: thirdparty/DBoW2/DBoW2/FORB.cpp:81-101 */
int orc_descriptor_distance(const uint8_t a[32], const uint8_t b[32])
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb; memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}
void orc_hamming_matrix(const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out)
{
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++)
            out[(size_t)i * nb + j] = (uint16_t)orc_descriptor_distance(a + 32 * (size_t)i, b + 32 * (size_t)j);
}
void orc_hamming_best2(const uint8_t* a, int na, const uint8_t* b, int nb,
                       int32_t* best_idx, uint16_t* best, uint16_t* second)
{
    for (int i = 0; i < na; i++) {
        int b1 = 0x7fffffff, b2 = 0x7fffffff, bi = -1;   /* INT_MAX like ORBmatcher.cc:438-440 */
        for (int j = 0; j < nb; j++) {
            int d = orc_descriptor_distance(a + 32 * (size_t)i, b + 32 * (size_t)j);
            if (d < b1) { b2 = b1; b1 = d; bi = j; }
            else if (d < b2) b2 = d;
        }
        best_idx[i] = bi;
        best[i] = (uint16_t)(b1 > 65535 ? 65535 : b1);
        second[i] = (uint16_t)(b2 > 65535 ? 65535 : b2);
    }
}

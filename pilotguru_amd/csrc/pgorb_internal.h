// pgorb_internal.h -- shared host/device declarations of libpgorb (gfx950 only).
//
// Data layout in HBM (all per context, sized for max_batch frames):
//   pyramid arena   level l, frame f : u8 plane, row pitch = align64(w_l), at
//                   lvl[l].img + f * lvl[l].fstride          (level 0 may alias the caller's
//                   device buffer when it is 4-byte aligned -- no copy)
//   cell slots      u32 cellCand[f][l][cell][cellCap_l] + i32 cellCount[f][cell]: K2 output,
//                   packed x | y<<12 | score<<24 (region-relative coordinates, i.e. pixel - 16,
//                   like vToDistributeKeys ORBextractor.cc:818-826)
//   key records     uint2 keys[f][l][candCap_l] = {packed candidate, node position | quadrant<<28}:
//                   K2's records compacted by K3's prologue + the quadtree's per-key state
//   selection       u32 sel[f][l][selCap]      quadtree result in the reference's output order
//   counters        i32 candCount[f][l], kpCount[f][l], status word
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pgorb.h"

#define PG_EDGE 16            // minBorderX = EDGE_THRESHOLD-3 (ORBextractor.cc:773)
#define PG_MAXL PGORB_MAX_LEVELS

struct PgQuadTab {            // column taps of 4 adjacent destination pixels (pyramid fast path)
    int32_t  base_dw;         // first aligned source dword
    uint32_t offs;            // 4 bits per pixel: byte offset of tap 0 from 4*base_dw (0..8)
    int16_t  a0[4], a1[4];    // 11-bit coefficients of tap 0 / tap 1
};

struct PgQuadTab2 {           // pyramid 4x4 fast path: one unaligned 8-byte window per quad and row
    int32_t  xb;              // byte offset of the window in a source row (<= sw - 8)
    uint32_t sel[4];          // v_perm_b32 selectors: tap0 -> bits 0..7, tap1 -> bits 16..23
    uint32_t coef[4];         // a0 | a1 << 16 (11-bit coefficients)
};

struct PgRowGrp {             // pyramid 4x4 fast path: everything a wave needs about its 4 destination rows
    int32_t  sFirst;          // first source row (yofs of the group's first row)
    uint32_t yrel4;           // the 4 row patterns, one byte each (bit 0: r0 offset, bit 1: r1 - r0)
    int16_t  ybeta[8];        // (b0, b1) of the 4 rows, 11-bit coefficients
    uint32_t pad[2];
};

struct PgLevel {
    // pyramid plane of this level
    uint8_t* img;             // frame 0 plane
    int64_t  fstride;         // bytes between consecutive frames' planes
    int32_t  w, h, pitch;
    // resize tables (level >= 1): per destination column / row (device pointers)
    const int32_t* xofs;      // [w]   source column of tap 0 ; tap 1 = xofs1
    const int32_t* xofs1;     // [w]   clamped second tap column
    const int16_t* xalpha;    // [2*w] 11-bit coefficients
    const int32_t* yofs;      // [2*h] clamped source rows of the two taps
    const int16_t* ybeta;     // [2*h]
    const PgQuadTab* qtab;    // [ceil(w/4)] or null when a quad spans more than 3 source dwords
    const PgQuadTab2* qtab2;  // [ceil(w/4)] or null when a quad's taps do not fit one 8-byte window
    const uint8_t* yrel;      // [h] row pattern of the 4x4 fast path, or null when it does not apply
    const PgRowGrp* rowgrp;   // [ceil(h/4)] the same, one 32-byte record per group of 4 rows (one scalar load)
    const int32_t* tilex;     // [ceil(w/256)] 16-aligned first source column of each 256-column tile (LDS-staged variant)
    int32_t  pyrCpr;          // 16-byte chunks per staged source row (0: the LDS-staged variant does not apply)
    int32_t  pyrRows;         // source rows staged per tile
    int32_t  pyrGpw;          // 4-row groups per wave: the tile is 256 columns x 16 * pyrGpw rows (2 = the 256 x 32 default;
                              // PGORB_PYR_TILE_ROWS = 16 | 32 | 64 at plan time, for the tile-size sweep in DESIGN.md)
    // cell grid (ORBextractor.cc:781-787)
    int32_t  nCols, nRows, wCell, hCell, cellBase;
    // quadtree (ORBextractor.cc:539-563)
    int32_t  quota, nIni, selCap;
    float    hX;
    // candidate storage
    int32_t  cellCap;         // candidate slots per cell = ceil(wCell/2)*ceil(hCell/2)
    int64_t  cellCandOff;     // u32 offset of this level's cell slots inside a frame's slab
    int32_t  candCap;
    int64_t  candOff;         // u32 offset of (frame 0, this level) inside a frame's slab
    int64_t  selOff;          // u32 offset inside a frame's selection slab
    int64_t  nodeOff;         // int offset of this level's node arrays inside a frame's node-scratch slab,
                              // or -1 when they fit the quadtree workgroup's LDS (the normal case)
    int32_t  nodeCap;
    int32_t  qtTabOff;        // K3's coordinate tables of this level inside PgPlan::qtTab
    float    scale;           // mvScaleFactor[level]
    float    patchSize;       // (float)(int)(31*scale)  ORBextractor.cc:836
};

// K3 -> K4-6: one record per selected keypoint, in K4-6's dispatch order; everything the descriptor wave needs about
// its keypoint AND its level, so that its prologue is two scalar round trips (arguments; record + the frame's counts)
struct PgSelRec {
    uint32_t cv;              // x | y << 12 | response << 24 (region-relative, like the candidate records)
    uint32_t posLevel;        // position in the reference's output list of the level | level << 16
    int32_t  pitch;           // level plane: row pitch
    uint32_t wh;              // w | h << 16
    uint64_t plane;           // address of THIS frame's plane of the level
    float    scale, patchSize;
};
static_assert(sizeof(PgSelRec) == 32, "one s_load_dwordx8");

// Fused resize + detect launch (fused.hip) with level l as the SOURCE: one wave per slot = bands (the level's cell rows + the edge bands
// above and below) x columns (the cell columns + the pseudo-cells along the left and right edge).  bands == null when the level does
// not take it (no next level, cells wider than 32 px, a generic scale factor).  Host-side only (not part of PgPlan, which travels to
// kernels by value).
struct PgFuseLevel {
    const int32_t* bands;     // [nBands] {first 4-row group of level + 1, groups}: band 0 = the rows above the first cell row, 1 .. nRows = the cell rows, then the rows below
    const int32_t* cols;      // [spb] {first quad of level + 1, quads, 65536 / quads + 1, 0}: entry s + 1 = slot column s (-1 = the left edge's pseudo-cell)
    int32_t  nBands, spb, rows;           // slots per band; staged rows per slot (hCell + 6)
};
struct PgFusePlan {
    PgFuseLevel lvl[PG_MAXL];
    int32_t  enabled;         // option "fused_levels": 1 = levels with fused tables take fused.hip's launch, 0 (default) = K1 + K2
};

#define PG_FAST_CPW_DEFAULT 1  // K2 cell records per wave
#define PG_QT_LEAF_CAP 4096   // >= the leaves of any count pyramid (quadtree.hip, QT_PYR_CAP)
struct PgPlan {
    PgLevel  lvl[PG_MAXL];
    int32_t  nlevels, totalCells, iniTh, minTh, tieMode;
    int32_t  selTotal;        // sum of selCap over levels (= per-frame keypoint bound)
    int64_t  candFrame;       // key records per frame in the cand arena
    int64_t  selFrame;        // selection records (PgSelRec, 32 B) per frame in the sel arena
    int64_t  nodeFrame;       // int per frame in node scratch
    int64_t  cellCandFrame;   // u32 per frame in the per-cell slot slab
    uint32_t* cellCand;       // K2 output: [frame][level][cell][cellCap]
    int32_t*  cellCount;      // K2 output: [frame][totalCells]
    // [totalCells] 64-byte record per cell, built with the plan (api.hip): everything K2 needs to
    // find its window in ONE scalar load -- the wave start used to be a chain of four dependent
    // scalar round trips (kernarg -> cell table -> level -> level fields).
    //   w0 level | cell index in the frame's cellCount array << 4      w1 iniX | iniY << 16
    //   w2 W | H << 8 | skip << 16 | cellCap << 17      w3 level pitch
    //   w4,w5 byte offset of (iniY, iniX - 1) in frame 0 of the level, relative to pyrBase
    //   w6 level frame stride                           w7 slot offset of the cell in the frame's slab
    //   w8-w15 validity of the necessary test's lanes and result bits for this cell's interior (lane masks, bit patterns)
    // Level 0 may alias the caller's buffer: its base / pitch / frame stride come from lvl[0].
    const uint32_t* cellTab;
    // the same records in K2's balanced dispatch order: [8 XCDs][cellsPerXcdBal], XCD x = the x-th eighth of every
    // level's cells; unused tail positions hold a padding record (cell index 0x0FFFFFFF): the wave returns at once
    const uint32_t* cellTabBal;
    int32_t  cellsPerXcdBal;
    int32_t  fastTilePitch;   // K2 tile-shape sweep: 0 = automatic, else the LDS window pitch in bytes (run-time-pitch instantiation)
    int32_t  fastWpb;         // K2 waves (= independent cells) per workgroup: 1 (default) or 4
    int32_t  fastCpw;         // K2 consecutive cell records per wave (option "fast_cells_per_wave"): PG_FAST_CPW_DEFAULT
    const uint8_t*  pyrBase;  // pyramid arena
    uint32_t* cand;           // K3: dense uint2 key records (2 u32 per key)
    uint32_t* sel;
    int32_t*  nodeScratch;
    int32_t*  candCount;      // [frame][PG_MAXL]
    int32_t*  kpCount;        // [frame][PG_MAXL]
    // K3 in two launches (round 4, quadtree.hip): the candidate pass runs on many small workgroups (k_qt_leaves) and leaves, per
    // (frame, level), the count and the best candidate of every depth-D descendant of the roots here; k_quadtree reads them
    const uint2* qtTab;       // per level: [regionW] {leaf column, rank part}, [regionH] {leaf row, rank part}, [2^D + 1] {first y of leaf row r, 0} (built with the plan)
    uint2*    qtLeaf;         // [frame][nlevels][PG_QT_LEAF_CAP] {count, (response << 24) | (0xFFFFFF - rank)}: every leaf written by k_qt_leaves in every batch
    int32_t   qtSplit;        // option "quadtree_split": 0 = the pass inside k_quadtree, 1 = two launches, 2 = chosen per launch (default; pg_launch_quadtree_levels)
    int32_t   qtThreads;      // option "quadtree_threads": 0 = chosen per launch (default), 256 | 512 | 1024 = that many threads per K3 workgroup
    int32_t   qtWide;         // 1 (default): 512-thread launches limited to two workgroups per CU by their LDS use the 128-VGPR instantiation (PGORB_QT_WIDE=0: never)
    int32_t*  status;         // device status word
};

// K3's count pyramid: depth for a level with nIni roots = the largest D <= 5 whose pyramid (nIni * (4^(D+1)-1)/3 counters)
// fits QT_PYR_CAP ints of LDS
#define QT_PYR_CAP 4096
__host__ __device__ __forceinline__ int qt_pyr_off(int nIni, int d) { return nIni * (((1 << (2 * d)) - 1) / 3); }
__host__ __device__ __forceinline__ int qt_pyr_depth(int nIni)
{
    int D = 5;
    while (D > 0 && qt_pyr_off(nIni, D + 1) > QT_PYR_CAP) D--;
    return D;
}
// One entry of K3's coordinate tables (quadtree.hip, the candidate pass): for region column c (isY = false) or row c (isY = true),
// as K2 stores coordinates, {its bits of the depth-D descendant index, its part of the candidate-order rank}.  The same text builds
// the tables with the plan on the host (api.hip) and inside k_quadtree on the device: single IEEE operations, nothing to contract.
#if defined(__HIP_DEVICE_COMPILE__)
#define PG_FDIV_RN(a, b) __fdiv_rn(a, b)
#define PG_FMUL_RN(a, b) __fmul_rn(a, b)
#else
#define PG_FDIV_RN(a, b) ((a) / (b))          /* host: one IEEE single-precision operation either way */
#define PG_FMUL_RN(a, b) ((a) * (b))
#endif
__host__ __device__ inline uint2 qt_tab_entry(bool isY, int c, float hX, int nIni, int regionH, int D, int wCell, int hCell, int nCols)
{
    const uint32_t mW = ((1u << 20) + wCell - 1) / wCell, mH = ((1u << 20) + hCell - 1) / hCell;
    int part, a0, a1;
    if (isY) { part = 0; a0 = 0; a1 = regionH; }
    else {
        int r = nIni >= 1 ? (int)PG_FDIV_RN((float)c, hX) : 0; // vpIniNodes[kp.pt.x/hX]  (:569)   (no root: the table is never read)
        const int top = (nIni > 1 ? nIni : 1) - 1;
        r = r < 0 ? 0 : (r > top ? top : r);
        part = r; a0 = (int)PG_FMUL_RN(hX, (float)r); a1 = (int)PG_FMUL_RN(hX, (float)(r + 1));
    }
    for (int d = 0; d < D; d++) {
        const int mid = a0 + ((a1 - a0 + 1) >> 1);
        const bool q = c >= mid;
        part = part * 4 + (q ? 1 : 0);
        a0 = q ? mid : a0; a1 = q ? a1 : mid;
    }
    const uint32_t cm = (uint32_t)(c - 3 > 0 ? c - 3 : 0);    // (order_rank's x - 3 / y - 3; K2's coordinates start at 3)
    if (isY) { const uint32_t ci = (cm * mH) >> 20; return make_uint2((uint32_t)part << 1, (ci * nCols * hCell + (cm - ci * hCell)) * wCell); }
    const uint32_t cj = (cm * mW) >> 20;
    return make_uint2((uint32_t)part, cj * hCell * wCell + (cm - cj * wCell));
}

// K3's candidate pass as its own launch (k_qt_leaves): a workgroup OWNS `rowsPer` of the 2^D leaf rows of a (frame, level) problem
// (a leaf row = the depth-D descendants with the same y path) and walks the cell rows that overlap them -- about QTP_CELLS cells
#define QTP_CELLS 256
__host__ __device__ __forceinline__ int qt_pass_rows_per_group(int ncells, int D)
{
    const int rows = 1 << D;
    int per = (rows * QTP_CELLS + (ncells > 1 ? ncells : 1) - 1) / (ncells > 1 ? ncells : 1);
    per = per < 1 ? 1 : (per > rows ? rows : per);
    return per;
}
__host__ __device__ __forceinline__ int qt_pass_groups(int ncells, int D) { const int per = qt_pass_rows_per_group(ncells, D); return ((1 << D) + per - 1) / per; }
__host__ __device__ __forceinline__ uint32_t qt_spread_bits(uint32_t v) { uint32_t r = 0; for (int k = 0; k < 6; k++) r |= ((v >> k) & 1u) << (2 * k); return r; }     // Morton: bit k -> bit 2k
__host__ __device__ __forceinline__ uint32_t qt_compact_bits(uint32_t v) { uint32_t r = 0; for (int k = 0; k < 6; k++) r |= ((v >> (2 * k)) & 1u) << k; return r; }

// MapPoint::PredictScale's logarithm (thirdparty/orb-slam2/src/MapPoint.cc:524 and Frame.cc:188: std::log(float), the platform's
// logf) under the parity contract shared with oracle/match_oracle.c (orc_log_f): a fixed double-precision sequence, rounded
// once to float -- x = m 2^e, m in (sqrt(1/2), sqrt(2)], s = (m - 1) / (m + 1), log x = e ln2 + 2 s (1 + s^2/3 + ... + s^20/21), no FMA.
__host__ __device__ inline float pg_log_f(float xf)
{
#ifdef __HIP_DEVICE_COMPILE__
#define PGL_M(a, b) __dmul_rn((a), (b))
#define PGL_A(a, b) __dadd_rn((a), (b))
#define PGL_D(a, b) __ddiv_rn((a), (b))
#else
#define PGL_M(a, b) ((a) * (b))
#define PGL_A(a, b) ((a) + (b))
#define PGL_D(a, b) ((a) / (b))
#endif
    if (xf != xf) return xf;
    if (!(xf > 0.0f)) return -__builtin_huge_valf();
    if (xf == __builtin_huge_valf()) return xf;
    double d = (double)xf;                                          // every positive float is a normal double
    uint64_t u = __builtin_bit_cast(uint64_t, d);
    int e = (int)((u >> 52) & 0x7FF) - 1023;
    u = (u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = __builtin_bit_cast(double, u);                        // [1, 2)
    if (m > 1.4142135623730951) { m = PGL_M(m, 0.5); e += 1; }
    const double s = PGL_D(PGL_A(m, -1.0), PGL_A(m, 1.0)), z = PGL_M(s, s);
    double p = 1.0 / 21.0;
    p = PGL_A(PGL_M(p, z), 1.0 / 19.0); p = PGL_A(PGL_M(p, z), 1.0 / 17.0); p = PGL_A(PGL_M(p, z), 1.0 / 15.0);
    p = PGL_A(PGL_M(p, z), 1.0 / 13.0); p = PGL_A(PGL_M(p, z), 1.0 / 11.0); p = PGL_A(PGL_M(p, z), 1.0 / 9.0);
    p = PGL_A(PGL_M(p, z), 1.0 / 7.0);  p = PGL_A(PGL_M(p, z), 1.0 / 5.0);  p = PGL_A(PGL_M(p, z), 1.0 / 3.0);
    p = PGL_A(PGL_M(p, z), 1.0);
    return (float)PGL_A(PGL_M((double)e, 0.6931471805599453), PGL_M(PGL_M(2.0, s), p));
#undef PGL_M
#undef PGL_A
#undef PGL_D
}
// MapPoint::PredictScale(currentDist, Frame*) (MapPoint.cc:516-531); (int)ceil(...) of a NaN / out-of-range value is what
// x86-64's cvttss2si returns, INT_MIN, i.e. level 0 after the clamp
__host__ __device__ inline int pg_predict_scale(float maxDistance, float currentDist, float logScaleFactor, int nlevels)
{
#ifdef __HIP_DEVICE_COMPILE__
    const float ratio = __fdiv_rn(maxDistance, currentDist);
    const float q = ceilf(__fdiv_rn(pg_log_f(ratio), logScaleFactor));
#else
    const float ratio = maxDistance / currentDist;
    const float q = ceilf(pg_log_f(ratio) / logScaleFactor);
#endif
    int nScale = (q != q || q >= 2147483648.0f || q < -2147483648.0f) ? (-2147483647 - 1) : (int)q;
    if (nScale < 0) nScale = 0;
    else if (nScale >= nlevels) nScale = nlevels - 1;
    return nScale;
}

// kernel launchers (each in its own .hip file)
void pg_launch_copy_level0(const PgPlan& P, const uint8_t* src, int stride, int64_t fstride,
                           int nframes, hipStream_t s);
void pg_launch_ingest(const PgPlan& P, const uint8_t* src, int stride, int64_t fstride, int sw, int sh, int channels,
                      int rgb_order, int rot, int vflip, int hflip, int nframes, hipStream_t s);
void pg_launch_color_to_gray(const PgPlan& P, const uint8_t* src, int stride, int64_t fstride, int channels,
                             int rgb_order, int nframes, hipStream_t s);
bool pg_launch_pyramid_level(const PgPlan& P, int level, int nframes, hipStream_t s, int32_t* clearWord = nullptr);
bool pg_launch_pyr_fast(const PgPlan& P, const PgFusePlan& F, int level, int nframes, hipStream_t s);          // fused.hip
void pg_launch_fast(const PgPlan& P, int nframes, hipStream_t s);
void pg_launch_fast_levels(const PgPlan& P, int nframes, int levelBeg, int levelEnd, hipStream_t s);
void pg_launch_quadtree(const PgPlan& P, int nframes, hipStream_t s);
void pg_launch_quadtree_levels(const PgPlan& P, int nframes, int levelBeg, int levelEnd, hipStream_t s);
void pg_launch_describe(const PgPlan& P, int nframes, pgorb_keypoint* d_kps, uint8_t* d_desc,
                        int cap_per_frame, int32_t* d_n, hipStream_t s);
void pg_launch_describe_levels(const PgPlan& P, int nframes, pgorb_keypoint* d_kps, uint8_t* d_desc,
                               int cap_per_frame, int32_t* d_n, int levelBeg, int levelEnd, hipStream_t s);
void pg_launch_hamming_matrix(const uint8_t* d_a, int na, const uint8_t* d_b, int nb,
                              uint16_t* d_out, hipStream_t s);
void pg_launch_prev_matched_init(const pgorb_keypoint* d_kps, int64_t rows, float* d_out, hipStream_t s);   // frame.hip
// matcher settings of ONE context (pgorb_set_option "matcher" / "match_mode"; defaults from the environment at pgorb_create)
struct PgMatchOpts {
    int popcount = 0;         // 1: the v_bcnt kernels for every size (PGORB_MATCH_POPCOUNT)
    int mode = -1;            // -1 = by grid size, 0 | 1 | 2 forced (PGORB_MATCH_MODE; match.hip)
};
PgMatchOpts pg_match_default_opts();      // what the environment says
size_t pg_match_scratch_bytes(const PgMatchOpts& o, int nb_max, int npairs);
void pg_launch_match_batch(const PgMatchOpts& o, const uint8_t* d_desc, const int32_t* d_n, int cap_per_frame,
                           const int32_t* d_pq, const int32_t* d_pt, int npairs, uint8_t* d_scratch,
                           int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second, hipStream_t s);
void pg_launch_best2(const PgMatchOpts& o, const uint8_t* d_a, int na, const uint8_t* d_b, int nb, uint8_t* d_scratch,
                     int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second, hipStream_t s);
bool pg_match_uses_popcount(const PgMatchOpts& o, int cap_per_frame);

static_assert(sizeof(PgPlan) <= 4000, "PgPlan is passed by value as a kernel argument (4 KiB limit)");

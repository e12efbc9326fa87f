// pgorb_internal.h -- shared host/device declarations of libpgorb (gfx950 only).
//
// Data layout in HBM (all per context, sized for max_batch frames):
//   pyramid arena   level l, frame f : u8 plane, row pitch = align64(w_l), at
//                   lvl[l].img + f * lvl[l].fstride          (level 0 may alias the caller's
//                   device buffer when it is 4-byte aligned -- no copy)
//   candidate arena u32 cand[f][l][candCap_l]  packed x | y<<12 | score<<24 (region-relative
//                   coordinates, i.e. pixel - 16, like vToDistributeKeys ORBextractor.cc:818-826)
//   key scratch     u32 kpos[f][l][candCap_l]  quadtree: current node position of every key
//   selection       u32 sel[f][l][selCap]      quadtree result in the reference's output order
//   counters        i32 candCount[f][l], kpCount[f][l], status word
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pgorb.h"

#define PG_EDGE 16            // minBorderX = EDGE_THRESHOLD-3 (ORBextractor.cc:773)
#define PG_MAXL PGORB_MAX_LEVELS

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
    // cell grid (ORBextractor.cc:781-787)
    int32_t  nCols, nRows, wCell, hCell, cellBase;
    // quadtree (ORBextractor.cc:539-563)
    int32_t  quota, nIni, selCap;
    float    hX;
    // candidate storage
    int32_t  candCap;
    int64_t  candOff;         // u32 offset of (frame 0, this level) inside a frame's slab
    int64_t  selOff;          // u32 offset inside a frame's selection slab
    int64_t  nodeOff;         // int offset inside a (frame) node-scratch slab
    int32_t  nodeCap;
    float    scale;           // mvScaleFactor[level]
    float    patchSize;       // (float)(int)(31*scale)  ORBextractor.cc:836
};

struct PgPlan {
    PgLevel  lvl[PG_MAXL];
    int32_t  nlevels, totalCells, iniTh, minTh, tieMode;
    int32_t  selTotal;        // sum of selCap over levels (= per-frame keypoint bound)
    int64_t  candFrame;       // u32 per frame in cand / kpos arenas
    int64_t  selFrame;        // u32 per frame in sel arena
    int64_t  nodeFrame;       // int per frame in node scratch
    uint32_t* cand;
    uint32_t* kpos;
    uint32_t* sel;
    int32_t*  nodeScratch;
    int32_t*  candCount;      // [frame][PG_MAXL]
    int32_t*  kpCount;        // [frame][PG_MAXL]
    int32_t*  status;         // device status word
};

// kernel launchers (each in its own .hip file)
void pg_launch_copy_level0(const PgPlan& P, const uint8_t* src, int stride, int64_t fstride,
                           int nframes, hipStream_t s);
void pg_launch_pyramid_level(const PgPlan& P, int level, int nframes, hipStream_t s);
void pg_launch_fast(const PgPlan& P, int nframes, hipStream_t s);
void pg_launch_quadtree(const PgPlan& P, int nframes, hipStream_t s);
void pg_launch_describe(const PgPlan& P, int nframes, pgorb_keypoint* d_kps, uint8_t* d_desc,
                        int cap_per_frame, int32_t* d_n, hipStream_t s);
void pg_launch_hamming_matrix(const uint8_t* d_a, int na, const uint8_t* d_b, int nb,
                              uint16_t* d_out, hipStream_t s);
void pg_launch_match_batch(const uint8_t* d_desc, const int32_t* d_n, int cap_per_frame,
                           const int32_t* d_pq, const int32_t* d_pt, int npairs,
                           int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second,
                           hipStream_t s);
void pg_launch_best2(const uint8_t* d_a, int na, const uint8_t* d_b, int nb,
                     int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second, hipStream_t s);

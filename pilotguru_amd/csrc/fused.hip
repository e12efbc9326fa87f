// fused.hip -- K1+K2 in one launch per pyramid level (round 6): every level is read from HBM ONCE.
// NOT the default path: bit-exact, and slower than K1 + K2 in every form measured (profiles/r06_fused_forms.txt: this form 1.07 ms
// for the seven launches against 0.86) -- neither K1 nor K2 is limited by the bytes it reads, and one wave that resizes AND detects
// queues both behind the same start-up latency.  pgorb_set_option(ctx, "fused_levels", 1) selects it.
//
// The reference builds the whole pyramid first (ORBextractor::ComputePyramid, thirdparty/orb-slam2/src/ORBextractor.cc:1106-1131)
// and then walks every level's 30-px cells with cv::FAST (ComputeKeyPointsOctTree, :765-829).  As two launches per level that is
// two trips to HBM for the same bytes: K1 reads level l to write level l+1, K2 reads level l again to detect (its staging-only
// build was 0.326 of its 0.585 ms, profiles/r05_k2_stages.txt).  Here the launch that resizes level l -> l+1 also detects level l:
//
//   slot   = one cell of level l -- or, along the frame's four edges, a pseudo-cell that only serves the resize: hCell + 6 source
//            rows from row 16 + i hCell (the cell window with its 3-px ring; band 0: from row 0) x 48 bytes from column
//            15 + j wCell (byte 0 = the cell's iniX - 1: K2's layout, interior column 0 on a dword boundary), staged ONCE by
//            LDS-DMA in 16-byte chunks from byte-unaligned global addresses.  ONE 64-lane wave per slot, one-wave workgroups: K2's
//            shape, K2's 4.9 KB of LDS per wave, eight waves per SIMD (a workgroup of four waves per TILE of cells amortised the
//            staging but held every cell's window for the workgroup's whole life: 16 waves per CU, 1.33 ms for the seven launches
//            against K1 + K2's 0.86 -- profiles/r06_fused_forms.txt, tools/experiments/r6_fused_cooperative_workgroup.hip.txt);
//   resize   the slot owns the 4-row groups of level l+1 whose first source row falls into its band and the quads whose first
//            tap falls into its columns (both partitions are exact: every destination pixel is written by exactly one slot):
//            lane = (quad, group), at most 8 x 8, each lane a 4 x 4 destination block from the slot's rows -- a quad's 8-byte
//            source window lies inside ONE slot (its offset there is < wCell <= 32); the arithmetic is pyramid.hip's
//            (cv::resize INTER_LINEAR, 11-bit fixed point), the two source rows of a destination row picked per lane;
//   detect   fast_cell.inc's detector on the slot -- the same code, the same per-cell semantics as k_fast_cells (cell interiors
//            tile the plane, NMS is window-local, the threshold falls back per cell).
// Everything the wave needs about its slot follows from its position and the kernel arguments: no cell record, no table in
// front of the DMA (k_fast_cells: kernarg -> record -> window, three dependent round trips; here two).  The last level is
// detected by k_fast_cells (fast.hip) as before.  Per level: algorithmic bytes w_l*h_l read + w_{l+1}*h_{l+1} written.
// A slot row is read 48 bytes wide, past the window into the neighbouring cell or the next row of the level: always inside the
// plane (the arena keeps 64 bytes of slack behind every plane) -- except when level 0 IS the caller's buffer, whose last row may
// end its allocation: there (safeLastRow) the chunks of the last row that would pass the row pitch are copied byte by byte.
//
// blockIdx -> slot is XCD-aware like K2's: XCD k takes the k-th eighth of the frame's slots in (band, column) order, so
// neighbours -- which share 128-B lines and 6 halo rows -- read them through the same L2.
#include "pgorb_internal.h"
#include <algorithm>
#include "fast_cell.inc"
#include "pyramid_rows4.inc"

extern __shared__ __attribute__((aligned(16))) uint8_t pg_fuse_smem[];
typedef uint32_t pg_u32x16 __attribute__((ext_vector_type(16)));

struct PgFuseArgs {
    const uint8_t* src;  int64_t sfstride;  uint8_t* dst;  int64_t dfstride;
    const PgQuadTab2* qtab;  const PgRowGrp* rowgrp;  const int32_t* bandTab;  const int32_t* colTab;
    int32_t* cellCount;  uint32_t* cellCand;  int32_t* status;
    int32_t spitch, sw, sh, dpitch, dh, nBands, spb, rows, nCols, nRows, wCell, hCell, cellBase, cellCap, totalCells, safeLastRow;
    uint32_t cellCandFrame, cellCandOff, spbMagic;
    int32_t iniTh, minTh, mapRows;
};

// The validity words of a cell's necessary test (fast_cell.inc, PgCellValid; the same arithmetic as PgPlan::cellTab w8-w15 in
// api.hip) from the interior's size -- all scalar: a slot's cell is known from the slot's position, no record is loaded
__device__ __forceinline__ PgCellValid pg_cell_valid(int IW, int IH)
{
    const int qFull = IW >> 2, rem = IW & 3, base = IH >> 3, rr = IH & 7;
    const int kLo = (rr + 1) >> 1, kHi = rr >> 1;                      // lanes whose row takes one step more than IH / 8
    PgCellValid V;
    V.qLt = ((1u << qFull) - 1u) * 0x01010101u;
    V.qEq = qFull < 8 ? (1u << qFull) * 0x01010101u : 0u;
    V.partial = (1u << (8 * rem)) - 1u;
    V.rowLo = kLo >= 4 ? 0xFFFFFFFFu : (1u << (8 * kLo)) - 1u;
    V.rowHi = (1u << (8 * kHi)) - 1u;
    V.stepBase = ((1u << min(base, 4)) - 1u) * 0x11111111u;
    V.stepMore = base < 4 ? 0x11111111u << base : 0u;
    V.fifth = base >= 5 ? 2u : (base == 4 ? 1u : 0u);
    return V;
}

// HResizeLinear of one staged source row for a lane's quad: (horizontal sum >> 4) << 4 of its four pixels (pyramid_rows4.inc, pyr_group)
__device__ __forceinline__ void fuse_hrow(const uint8_t* rowp, uint32_t sh3, const PgQuadTab2& T, uint32_t (&H)[4])
{
    const uint32_t* d = reinterpret_cast<const uint32_t*>(rowp);
    const uint32_t d0 = d[0], d1 = d[1], d2 = d[2];
    const uint32_t wx = __builtin_amdgcn_alignbyte(d1, d0, sh3), wy = __builtin_amdgcn_alignbyte(d2, d1, sh3);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t taps = __builtin_amdgcn_perm(wy, wx, T.sel[j]);                      // tap0 | tap1 << 16
        H[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(pg_us2, taps), __builtin_bit_cast(pg_us2, T.coef[j]), 0u, false) & ~15u;
    }
}

template <int DUMMY>
__global__ __launch_bounds__(64, 8) void k_fast_resize(const PgFuseArgs A)
{
    const int lane = threadIdx.x;
    const uint32_t frame = blockIdx.y;
    // slot t of the frame in (band, column) order; band b: 0 = the rows above the first cell row, 1 .. nRows = the cell rows, beyond
    // = the rows below the last one; column s: -1 = the left edge's pseudo-cell (source columns from 0), s >= 0 = from column
    // 15 + s wCell (cell column s; past nCols: pseudo-cells)
    const int t = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (t >= A.nBands * A.spb) return;
    const int b = (int)__umulhi((uint32_t)t, A.spbMagic), s = t - b * A.spb - 1;
    const int Y = b ? PG_EDGE + (b - 1) * A.hCell : 0;
    const int sx = s < 0 ? 0 : PG_EDGE - 1 + s * A.wCell;
    uint8_t* tile = pg_fuse_smem;                                      // [rows][48]
    uint8_t* smap = tile + A.rows * 48;                                // [mapRows][40], 1-px zero rim
    uint16_t* list = reinterpret_cast<uint16_t*>(smap + A.mapRows * 40);
    {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);                    // the score map, BEFORE the window loads (behind them every ds_write would wait
        reinterpret_cast<uint4*>(smap)[lane] = z;                      // for the DMA); two whole steps of 1 KiB: what they clear past the map is the list
        reinterpret_cast<uint4*>(smap)[lane + 64] = z;
    }
    // (1) stage the slot: 3 chunks per row, 21 rows per instruction
    {
        const uint8_t* fb = A.src + (int64_t)frame * A.sfstride;
        const int r0 = (lane * 21846) >> 16, ch = lane - 3 * r0;       // lane / 3, lane % 3
        const int gx = sx + ch * 16;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k * 21 < A.rows) {                                     // wave-uniform
                const int row = r0 + k * 21;
                const int grow = min(Y + row, A.sh - 1);
                if (lane < 63 && row < A.rows) {
                    const uint8_t* g = fb + (int64_t)grow * A.spitch + gx;
                    if (A.safeLastRow && grow == A.sh - 1 && gx + 16 > A.spitch) {
                        for (int i = 0; i < 16 && gx + i < A.spitch; i++) tile[row * 48 + ch * 16 + i] = g[i];
                    } else {
                        __builtin_amdgcn_global_load_lds((pg_gptr_t)g, (pg_lptr_t)(tile + k * 21 * 48), 16, 0, 0);
                    }
                }
            }
        }
    }
    // ... and, under the DMA, the lane's share of the resize: quad ql of the column's nq, group gl of the band's ng
    const int gBeg = A.bandTab[2 * b], ng = A.bandTab[2 * b + 1];
    const int qBeg = A.colTab[4 * (s + 1)], nq = A.colTab[4 * (s + 1) + 1], qMagic = A.colTab[4 * (s + 1) + 2];
    const int gl = (lane * qMagic) >> 16, ql = lane - gl * nq;         // lane / nq, lane % nq (exact for lane < 64: api.hip checks)
    const bool on = gl < ng;
    const int quad = qBeg + (on ? ql : 0), grp = gBeg + (on ? gl : 0);
    const PgQuadTab2 T = A.qtab[quad];
    const uint32_t* rw = reinterpret_cast<const uint32_t*>(A.rowgrp + grp);
    const uint32_t rS = rw[0], rY = rw[1], rB0 = rw[2], rB1 = rw[3], rB2 = rw[4], rB3 = rw[5];
    __builtin_amdgcn_s_waitcnt(0);                                     // vmcnt(0): the DMA has landed
    PG_WAVE_SYNC();
    // (2) resize: a 4 x 4 destination block per lane.  Destination row D takes source rows rel + D + e0 and that + e1 (e0, e1 in
    // {0, 1}: PgRowGrp::yrel4) -- per LANE here (the lanes of a wave hold different groups), so each row is resized horizontally
    // where it is needed instead of six shared rows and a wave-uniform choice among them
    if (on && nq > 0) {
        const int o = T.xb - sx;                                       // window offset in a slot row
        const uint8_t* lb = tile + (o & ~3);
        const uint32_t sh3 = (uint32_t)(o & 3);
        const int rel = (int)rS - Y;
        uint8_t* dp = A.dst + (int64_t)frame * A.dfstride + (int64_t)(4 * grp) * A.dpitch + quad * 4;
        const uint32_t bw[4] = {rB0, rB1, rB2, rB3};
#pragma unroll
        for (int D = 0; D < 4; D++) {
            const int f = (int)(rY >> (8 * D)) & 3;
            const int rA = min(rel + D + (f & 1), A.rows - 1), rBq = min(rA + (f >> 1), A.rows - 1);
            uint32_t HA[4], HB[4];
            fuse_hrow(lb + rA * 48, sh3, T, HA);
            fuse_hrow(lb + rBq * 48, sh3, T, HB);
            const uint32_t b0s = (bw[D] & 0xFFFFu) << 12, b1s = (bw[D] >> 16) << 12;
            uint32_t out = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t v = (pg_mulhi_u24(HA[j], b0s) + pg_mulhi_u24(HB[j], b1s) + 2u) >> 2;   // VResizeLinear, <= 255
                out |= v << (8 * j);
            }
            if (4 * grp + D < A.dh) *reinterpret_cast<uint32_t*>(dp + (int64_t)D * A.dpitch) = out;
        }
    }
    // (3) detect, when the slot is a cell (ORBextractor.cc:791-806)
    const int cellRow = b - 1;
    if (b < 1 || b > A.nRows || s < 0 || s >= A.nCols) return;
    const int maxBorderX = A.sw - PG_EDGE, maxBorderY = A.sh - PG_EDGE;
    const int iniX = PG_EDGE + s * A.wCell, iniY = Y;
    const int W = min(iniX + A.wCell + 6, maxBorderX) - iniX, H = min(iniY + A.hCell + 6, maxBorderY) - iniY;
    const int cidx = cellRow * A.nCols + s;
    int32_t* cellCnt = A.cellCount + ((uint64_t)frame * (uint32_t)A.totalCells + (uint32_t)(A.cellBase + cidx));
    if (iniY >= maxBorderY - 3 || iniX >= maxBorderX - 6 || W < 7 || H < 7) {      // skipped cell (:794, :803) or a window cv::FAST finds nothing in
        if (lane == 0) *cellCnt = 0;
        return;
    }
    const int IW = W - 6, IH = H - 6;
    uint32_t* out = A.cellCand + ((uint64_t)frame * A.cellCandFrame + (A.cellCandOff + (uint32_t)cidx * (uint32_t)A.cellCap));
    const PgLaneValid valid = pg_lane_valid(pg_cell_valid(IW, IH));
    const int xoff = 3 + iniX - PG_EDGE, yoff = 3 + iniY - PG_EDGE;
    const int total = fast_cell_detect<48, 40, true>(A.status, tile, 48, smap, 40, A.mapRows, IW, IH, A.iniTh, A.minTh, list, out, A.cellCap,
                                                     xoff, yoff, lane, valid);
    if (lane == 0) *cellCnt = min(total, A.cellCap);
}

// Resize level `level` -> level + 1 and detect level `level` in one launch.  Returns false (nothing launched) when the level has no
// fused tables or the source pitch does not tile into 16-byte chunks (an aliased caller buffer): the caller then takes K1 + K2.
bool pg_launch_pyr_fast(const PgPlan& P, const PgFusePlan& FP, int level, int nframes, hipStream_t s)
{
    const PgLevel& S = P.lvl[level];
    const PgLevel& D = P.lvl[level + 1];
    const PgFuseLevel& F = FP.lvl[level];
    if (!F.bands) return false;
    PgFuseArgs A = {};
    A.src = S.img; A.sfstride = S.fstride; A.dst = D.img; A.dfstride = D.fstride;
    A.qtab = D.qtab2; A.rowgrp = D.rowgrp; A.bandTab = F.bands; A.colTab = F.cols;
    A.cellCount = P.cellCount; A.cellCand = P.cellCand; A.status = P.status;
    A.spitch = S.pitch; A.sw = S.w; A.sh = S.h; A.dpitch = D.pitch; A.dh = D.h;
    A.nBands = F.nBands; A.spb = F.spb; A.rows = F.rows; A.wCell = S.wCell; A.hCell = S.hCell;
    A.nCols = S.nCols; A.nRows = S.nRows; A.cellBase = S.cellBase; A.cellCap = S.cellCap; A.totalCells = P.totalCells;
    A.cellCandFrame = (uint32_t)P.cellCandFrame; A.cellCandOff = (uint32_t)S.cellCandOff;
    A.spbMagic = (uint32_t)(((1ull << 32) / (uint64_t)F.spb) + 1ull);
    // the caller's buffer as level 0 (no slack behind its last row): see the header
    A.safeLastRow = (level == 0 && S.img != P.pyrBase) ? 1 : 0;
    A.iniTh = P.iniTh; A.minTh = P.minTh;
    // window [rows][48] + score map + list, like k_fast_cells (the map is cleared in two whole 1-KiB steps)
    A.mapRows = S.hCell + 2;
    const size_t lds = (((size_t)F.rows * 48 + std::max((size_t)A.mapRows * 40 + FAST_LIST_CAP * 2, (size_t)2048) + 16) + 15) & ~(size_t)15;
    const int ns = F.nBands * F.spb;
    dim3 block(64), grid((ns + 7) & ~7, nframes);
    hipLaunchKernelGGL(k_fast_resize<0>, grid, block, lds, s, A);
    return true;
}

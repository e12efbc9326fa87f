// fused.hip -- K1+K2 in one launch per pyramid level (round 6): every level is read from HBM ONCE.
//
// The reference builds the whole pyramid first (ORBextractor::ComputePyramid, thirdparty/orb-slam2/src/ORBextractor.cc:1106-1131)
// and then walks every level's 30-px cells with cv::FAST (ComputeKeyPointsOctTree, :765-829).  As two launches per level that is
// two trips to HBM for the same bytes: K1 reads level l to write level l+1, K2 reads level l again to detect (its staging-only
// build was 0.326 of its 0.585 ms, profiles/r05_k2_stages.txt).  Here the launch that resizes level l -> l+1 also detects level l:
//
//   tile   = one CELL ROW of level l (hCell + 6 source rows: the cell windows with their 3-px rings) x NS cells across
//            (NS * wCell + 6 source columns), staged ONCE into LDS by LDS-DMA in aligned 16-byte chunks;
//   resize   the tile owns the 4-row groups of level l+1 whose first source row falls into the band and the quads whose
//            first tap falls into its columns (both partitions are exact: every destination pixel is written by exactly one
//            tile); the arithmetic is pyramid.hip's (pyr_group: cv::resize INTER_LINEAR, 11-bit fixed point);
//   detect   each of the workgroup's four waves takes cells of the tile: it cuts the cell's window out of the band into a
//            private 48-byte-pitch tile with interior column 0 on a dword boundary (two aligned dwords + v_alignbyte per
//            dword: unaligned LDS reads are served one lane at a time on this part, tools/ubench/lds_ring.hip) and runs
//            fast_cell.inc's detector on it -- the same code, the same per-cell semantics as k_fast_cells (cell interiors
//            tile the plane, NMS is window-local, the threshold falls back per cell).
// The bands above the first and below the last cell row (16 + 3 rows each) are tiles without cells; the last level is
// detected by k_fast_cells (fast.hip) as before.  Per level: algorithmic bytes w_l*h_l read + w_{l+1}*h_{l+1} written.
//
// blockIdx -> tile is XCD-aware like K1's: XCD k takes the k-th eighth of the frame's tiles in (band, column) order, so tiles
// that share halo rows / boundary chunks read them through the same L2.
#include "pgorb_internal.h"
#include <algorithm>
#include "fast_cell.inc"
#include "pyramid_rows4.inc"

extern __shared__ __attribute__((aligned(16))) uint8_t pg_fuse_smem[];
typedef uint32_t pg_u32x16 __attribute__((ext_vector_type(16)));

struct PgFuseArgs {
    const uint8_t* src;  int64_t sfstride;  uint8_t* dst;  int64_t dfstride;
    const PgQuadTab2* qtab;  const PgRowGrp* rowgrp;  const int32_t* bandTab;  const int32_t* colTab;
    const uint32_t* cellTab;                // this level's records (PgPlan::cellTab + 16 * cellBase)
    int32_t* cellCount;  uint32_t* cellCand;  int32_t* status;
    int32_t spitch, sh, dpitch, dw, dh, nTx, nBands, cpr, cprInv, rows, nCols, totalCells;
    uint32_t cellCandFrame, nTxMagic;
    int32_t iniTh, minTh, waveLds, bandBytes;
};

template <int DUMMY>
__global__ __launch_bounds__(256, 5) void k_pyr_fast(const PgFuseArgs A)
{
    const int nt = A.nTx * A.nBands;
    const int t = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (t >= nt) return;                                               // padding tile (whole workgroup)
    const int b = (A.nTx == 1) ? t : (int)__umulhi((uint32_t)t, A.nTxMagic), tx = t - b * A.nTx;
    const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const uint32_t frame = blockIdx.z;
    // band record {first staged source row, first group, end group, cell row or -1}; column record {first staged column (16-aligned),
    // first quad, end quad, first cell column, cells}
    const int32_t* br = A.bandTab + 4 * b;
    const int32_t* cr = A.colTab + 8 * tx;
    const int Y = br[0], gBeg = br[1], gEnd = br[2], cellRow = br[3];
    const int x0a = cr[0], qBeg = cr[1], qEnd = cr[2], cellCol0 = cr[3], nCells = cr[4];
    const int BP = A.cpr * 16;
    uint8_t* band = pg_fuse_smem;                                      // [rows][BP]
    // (1) stage the band: cpr lanes per row, 64 / cpr rows per instruction (pyramid.hip's staging)
    {
        const int rowsPer = 64 / A.cpr;
        const int r0 = (lane * A.cprInv) >> 16, ch = lane - r0 * A.cpr;
        const uint8_t* sb = A.src + (int64_t)frame * A.sfstride + x0a + ch * 16;
        const bool laneOn = r0 < rowsPer && x0a + ch * 16 + 16 <= A.spitch;      // never past the row pitch
        for (int k = wv; k * rowsPer < A.rows; k += 4) {
            const int r = k * rowsPer + r0;
            if (laneOn && r < A.rows)
                __builtin_amdgcn_global_load_lds((pg_gptr_t)(sb + (int64_t)min(Y + r, A.sh - 1) * A.spitch),
                                                 (pg_lptr_t)(band + k * rowsPer * BP), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    // (2) resize: the band's 4-row groups of the destination level, one per wave and trip; lane = quad
    {
        const int quad = qBeg + lane;
        if (quad < qEnd) {
            const PgQuadTab2 T = A.qtab[quad];
            const int o = T.xb - x0a;                                  // window offset in a staged row
            const uint8_t* lb = band + (o & ~3);
            const uint32_t sh3 = (uint32_t)(o & 3);
            uint8_t* dbase = A.dst + (int64_t)frame * A.dfstride + quad * 4;
            for (int g = gBeg + wv; g < gEnd; g += 4) {
                const uint32_t* rw = reinterpret_cast<const uint32_t*>(A.rowgrp + g);
                const uint32_t ra[8] = {rw[0], rw[1], rw[2], rw[3], rw[4], rw[5], rw[6], rw[7]};
                const PgRowGrp R = pyr_unpack_group(ra);
                const int rel = R.sFirst - Y;
                PgU2 w[6];
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const uint32_t* d = reinterpret_cast<const uint32_t*>(lb + min(rel + k, A.rows - 1) * BP);
                    const uint32_t d0 = d[0], d1 = d[1], d2 = d[2];
                    w[k].x = __builtin_amdgcn_alignbyte(d1, d0, sh3);
                    w[k].y = __builtin_amdgcn_alignbyte(d2, d1, sh3);
                }
                pyr_group(w, T, R, 4 * g, A.dh, dbase, A.dpitch);
            }
        }
    }
    // (3) detect: cells wv, wv + 4, ... of the tile
    if (cellRow < 0) return;
    uint8_t* tile = pg_fuse_smem + A.bandBytes + wv * A.waveLds;       // [rows][48], this wave's private window
    uint8_t* smap = tile + A.rows * 48;                                // [hCell + 2][40], 1-px zero rim
    uint16_t* list = reinterpret_cast<uint16_t*>(smap + (A.rows - 4) * 40);
    for (int ci = wv; ci < nCells; ci += 4) {
        if (ci != wv) PG_WAVE_SYNC();
        const int cidx = __builtin_amdgcn_readfirstlane(cellRow * A.nCols + cellCol0 + ci);
        const uint32_t* recp = A.cellTab + 16 * (int64_t)cidx;
        pg_u32x16 rec;                                                 // ONE scalar load of the cell's 64-byte record (PgPlan::cellTab)
        asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rec) : "s"(recp) : "memory");
        const uint32_t r0w = rec[0], r1w = rec[1], r2w = rec[2], r7w = rec[7];
        int32_t* cellCnt = A.cellCount + ((uint64_t)frame * (uint32_t)A.totalCells + (r0w >> 4));
        if (r2w & 0x10000u) {                                          // skipped cell (:794, :803)
            if (lane == 0) *cellCnt = 0;
            continue;
        }
        const int iniX = r1w & 0xFFFF, iniY = r1w >> 16;
        const int W = r2w & 0xFF, H = (r2w >> 8) & 0xFF, cellCap = r2w >> 17;
        const int IW = W - 6, IH = H - 6;
        // the window, 12 dwords per row, byte 0 = global column iniX - 1 (interior column 0 on byte 4)
        {
            const int bx = iniX - 1 - x0a;
            const uint32_t m = (uint32_t)(bx & 3);
            const uint8_t* sbase = band + (iniY - Y) * BP + (bx & ~3);
            uint32_t* t32 = reinterpret_cast<uint32_t*>(tile);
            for (int k = lane; k < H * 12; k += 64) {
                const int r = (k * 21846) >> 18, d = k - 12 * r;       // k / 12 for k < 1536
                const uint32_t* p = reinterpret_cast<const uint32_t*>(sbase + r * BP) + d;
                t32[k] = __builtin_amdgcn_alignbyte(p[1], p[0], m);
            }
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);                // the score map (two whole steps of 1 KiB; what they clear past it is the list)
            reinterpret_cast<uint4*>(smap)[lane] = z;
            reinterpret_cast<uint4*>(smap)[lane + 64] = z;
        }
        PG_WAVE_SYNC();
        uint32_t* out = A.cellCand + ((uint64_t)frame * A.cellCandFrame + r7w);
        const PgCellValid cv = {rec[8], rec[9], rec[10], rec[11], rec[12], rec[13], rec[14], rec[15]};
        const PgLaneValid valid = pg_lane_valid(cv);
        const int xoff = 3 + iniX - PG_EDGE, yoff = 3 + iniY - PG_EDGE;
        const int total = fast_cell_detect<48, 40, true>(A.status, tile, 48, smap, 40, A.rows - 4, IW, IH, A.iniTh, A.minTh, list, out, cellCap,
                                                         xoff, yoff, lane, valid);
        if (lane == 0) *cellCnt = min(total, cellCap);
    }
}

// Resize level `level` -> level + 1 and detect level `level` in one launch.  Returns false (nothing launched) when the level has no
// fused tables or the source pitch does not tile into 16-byte chunks (an aliased caller buffer): the caller then takes K1 + K2.
bool pg_launch_pyr_fast(const PgPlan& P, const PgFusePlan& FP, int level, int nframes, hipStream_t s)
{
    const PgLevel& S = P.lvl[level];
    const PgLevel& D = P.lvl[level + 1];
    const PgFuseLevel& F = FP.lvl[level];
    if (!F.bands || S.pitch % 16 != 0) return false;
    PgFuseArgs A = {};
    A.src = S.img; A.sfstride = S.fstride; A.dst = D.img; A.dfstride = D.fstride;
    A.qtab = D.qtab2; A.rowgrp = D.rowgrp; A.bandTab = F.bands; A.colTab = F.cols;
    A.cellTab = P.cellTab + 16 * (size_t)S.cellBase;
    A.cellCount = P.cellCount; A.cellCand = P.cellCand; A.status = P.status;
    A.spitch = S.pitch; A.sh = S.h; A.dpitch = D.pitch; A.dw = D.w; A.dh = D.h;
    A.nTx = F.nTx; A.nBands = F.nBands; A.cpr = F.cpr; A.cprInv = 65536 / F.cpr + 1; A.rows = F.rows;
    A.nCols = S.nCols; A.totalCells = P.totalCells; A.cellCandFrame = (uint32_t)P.cellCandFrame;
    A.nTxMagic = F.nTx > 1 ? (uint32_t)(((1ull << 32) / (uint64_t)F.nTx) + 1ull) : 0u;
    A.iniTh = P.iniTh; A.minTh = P.minTh;
    // per wave: window [rows][48] + score map + list, like k_fast_cells (the map is cleared in two whole 1-KiB steps)
    const int mapRows = F.rows - 4;                                // hCell + 2
    A.waveLds = (int)(((size_t)F.rows * 48 + std::max((size_t)mapRows * 40 + FAST_LIST_CAP * 2, (size_t)2048) + 16 + 15) & ~(size_t)15);
    A.bandBytes = F.rows * F.cpr * 16 + 16;                    // (+16: the cut reads one dword past a row's last chunk)
    const size_t lds = (size_t)A.bandBytes + 4 * (size_t)A.waveLds;
    const int nt = F.nTx * F.nBands;
    dim3 block(64, 4), grid((nt + 7) & ~7, 1, nframes);
    hipLaunchKernelGGL(k_pyr_fast<0>, grid, block, lds, s, A);
    return true;
}

// fast.hip -- K2: per-cell FAST-9/16 score + 3x3 NMS + per-cell threshold fallback (gfx950).
//
// Restates the cell loop of ORBextractor::ComputeKeyPointsOctTree
// (thirdparty/orb-slam2/src/ORBextractor.cc:765-829), i.e. thousands of small
// cv::FAST(window, iniThFAST, true) calls with a cv::FAST(window, minThFAST, true) retry for
// empty cells (:808-816), as ONE launch: one 64-lane wave per 30-px cell, all cells of all
// pyramid levels of all frames of the batch in one grid.  Like the reference the wave runs
// the detector at iniThFAST first and only re-runs the (much denser) minThFAST pass when the
// cell came back empty -- on textured frames almost every cell is settled by the cheap pass.
//
// One detector pass at threshold t == cv::FAST(window, t, nonmax=true):
//   score(p) = (best 9-arc minimum |centre - ring|) - 1 for FAST corners (independent of t),
//   0 otherwise; keep p iff score(p) > score of all 8 neighbours, where everything outside the
//   window interior counts as 0 (the reference's window-local score buffers).
//
// Structure per wave:
//  (1) the (wCell+6)x(hCell+6) window is staged into LDS by LDS-DMA (global_load_lds, 16 B per
//      lane, byte-unaligned source) so that interior column 0 sits on an LDS dword boundary;
//      the wave finds its window through one 64-byte cell record (a single scalar load);
//      (persistent waves that prefetch the next window measured slower twice -- with register
//      prefetch 2x, with LDS-DMA double buffering 1.5x: the kernel sits between a latency floor
//      and the issue rate of its eight waves per SIMD, no single resource binds it (DESIGN.md
//      section 6), and it needs those eight waves more than it needs one wave's latency hidden)
//  (2) each lane tests a QUAD of 4 horizontally adjacent pixels per step from 5 aligned LDS
//      dwords (centre, left, right, 3 rows up, 3 rows down), FOUR pixels per 32-bit operation
//      (quick_pass_b: v_lerp_u8 half-differences, bytes in place; the 16-bit-field form quick_pass
//      serves cells wider than 32 px):
//      a pixel can only be a corner if both opposite ring pairs (0,8) and (4,12) contain a
//      darker (or a brighter) pixel (minThFAST pass: all four pairs incl. the diagonals); the
//      few percent that pass are compacted into an LDS list with ONE DPP prefix sum per pass;
//  (3) exact 16-ring scores (min3/max3 sliding arcs) for the compacted list, all lanes busy;
//  (4) NMS on an LDS score map; survivors go to the cell's own fixed slot range of the
//      (frame, level) candidate slab plus a per-cell count -- no atomics (a single device-scope
//      counter per level serialised ~2000 cells at ~180 ns each).  A cell can hold at most
//      ceil(IW/2)*ceil(IH/2) survivors (no two are 8-adjacent), which is the slot count, so
//      nothing can overflow.  K3 compacts the slots; order inside a cell is arbitrary and
//      everything downstream orders by the reference's (cell row, cell col, y, x) rank.
//
// blockIdx -> cell mapping is XCD-aware: the dispatcher places consecutive workgroups on
// consecutive XCDs (private L2s), so cell c = (b % 8) * (G/8) + b / 8 gives every XCD a
// contiguous run of cells -- neighbouring cells share 128-B lines and 6 halo rows.
//
// Roofline: HBM-read bound in principle (one pass over all pyramid pixels, 1 B/px);
// algorithmic bytes per frame = sum_l w_l*h_l.
#include "pgorb_internal.h"
#include <stdlib.h>
#include <string.h>
#include "fast_cell.inc"

extern __shared__ __attribute__((aligned(16))) uint8_t pg_fast_smem[];


// TPC / MPC: compile-time tile and score-map pitches of the common geometry (cells up to 36 px:
// TP = 48, map pitch 40), so that ring / neighbour offsets are instruction immediates; 0 = use the
// run-time values (larger cells).
// NARROW: every cell interior of the plan is at most 32 px wide (8 quads per row): the 16-quad variant of the
// necessary test and its per-lane set-up are compiled out (true for 1080p / 2160p / 480p: wCell <= 32).
// WPB waves per workgroup, each wave an independent cell (no workgroup barrier anywhere: the waves only share the
// launch and the LDS allocation, 4 x fewer workgroups for the dispatcher).
// Everything a K2 wave reads from its argument block: 128 bytes = two s_load_dwordx16.  The wave (re)loads them at the start of
// EVERY cell it walks, by explicit scalar loads: nothing but the loop counter is alive from one cell to the next, so the cell body
// compiles like the one-cell kernel (kept in SGPRs across the body, the arguments cost ~20 spills into VGPR lanes and scratch,
// and the launch ran 36 % slower -- round 3 had met the same wall).
struct PgFastArgs {
    const uint8_t* l0img;  int64_t l0fstride;  const uint8_t* pyrBase;  int32_t* cellCount;                       // dwords 0-7
    uint32_t* cellCand;  int32_t* status;  const uint32_t* tab;  int32_t l0pitch, totalCells;                       // 8-15
    uint32_t cellCandFrame;  int32_t iniTh, minTh, TPr, tileRows, MPr, mapRows, cellsPerXcd;                        // 16-23
    int32_t chunkInv, cell0, cellEnd, waveLds, cpw, pad0, pad1, pad2;                                               // 24-31
};
static_assert(sizeof(PgFastArgs) == 128, "two s_load_dwordx16");
typedef uint32_t pg_u32x16 __attribute__((ext_vector_type(16)));

template <int TPC, int MPC, bool NARROW, int WPB>
__global__ __launch_bounds__(64 * WPB, 8) void k_fast_cells(const PgFastArgs KA)
{
    // Records [cell0, cellEnd) of `tab`.  The default table is in a BALANCED dispatch order (api.hip): XCD x -- the
    // dispatcher deals consecutive workgroups to consecutive XCDs -- gets the x-th eighth of EVERY level's cells, a
    // contiguous band per level, so neighbours still share L2 lines and every XCD sees the same mix of cheap cells
    // (level 0: ~15 candidates) and expensive ones (upper levels: 40..170 candidates, several score rounds).  In plain
    // cell order XCD 7 held only upper-level cells and the launch waited for it while XCDs 0-2 idled.
    // The canonical-order table serves one level per launch when the pyramid chain runs beside K2.
    // cpw CONSECUTIVE records per wave (round 5; option "fast_cells_per_wave"): a one-wave workgroup's slot stands empty for ~0.7 us
    // between two waves (SQ_WAVE_CYCLES against slots x time: 88 % occupancy at one cell per wave) -- a wave that walks several
    // records pays that once.  No prefetch: the cells run one after the other on the same LDS.
    auto one_cell = [&](const int j) -> int {
    // (the lane id through an opaque copy per cell: what depends on the lane alone is recomputed per cell instead of being hoisted
    //  out of the loop, where a dozen per-lane constants alive across the cell body push it into scratch)
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wv = (WPB > 1) ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
    // Two scalar round trips to the window address: (1) the argument block, including what level 0 needs when it aliases the
    // caller's buffer, (2) ONE s_load_dwordx16 of the cell's record (pgorb_internal.h, PgPlan::cellTab).
    pg_u32x16 a0, a1;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a0), "=&s"(a1) : "s"(__builtin_amdgcn_kernarg_segment_ptr()) : "memory");
#define PG_A64(T, v, k) reinterpret_cast<T>((uintptr_t)(((uint64_t)(v)[(k) + 1] << 32) | (v)[k]))
    const uint8_t* l0img = PG_A64(const uint8_t*, a0, 0);
    const int64_t l0fstride = (int64_t)(((uint64_t)a0[3] << 32) | a0[2]);
    const uint8_t* pyrBase = PG_A64(const uint8_t*, a0, 4);
    int32_t* const cellCountBase = PG_A64(int32_t*, a0, 6);
    uint32_t* const cellCandBase = PG_A64(uint32_t*, a0, 8);
    int32_t* const statusPtr = PG_A64(int32_t*, a0, 10);
    const uint32_t* tab = PG_A64(const uint32_t*, a0, 12);
#undef PG_A64
    const int l0pitch = (int)a0[14], totalCells = (int)a0[15];
    const uint32_t cellCandFrame = a1[0];                          // u32 slots per frame: < 2^32 (make_plan)
    const int iniTh = (int)a1[1], minTh = (int)a1[2], TPr = (int)a1[3], tileRows = (int)a1[4], MPr = (int)a1[5], mapRows = (int)a1[6];
    const int cellsPerXcd = (int)a1[7], chunkInv = (int)a1[8], cell0 = (int)a1[9], cellEnd = (int)a1[10], waveLds = (int)a1[11], cpw = (int)a1[12];
    const int TP = TPC ? TPC : TPr, mapPitch = MPC ? MPC : MPr;
    const uint32_t frame = blockIdx.y;                             // (unsigned: 32 x 32 -> 64-bit scalar multiplies, two instructions each)
    const int slot = ((int)(blockIdx.x >> 3) * WPB + wv) * cpw + j;           // within this XCD's run of records
    const int cell = cell0 + (int)(blockIdx.x & 7) * cellsPerXcd + slot;      // position in the table
    const uint32_t* recp = tab + 16 * (int64_t)cell;               // (the tables end in 4 x 64 + 8 records of slack: api.hip)
    pg_u32x16 rec;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rec) : "s"(recp) : "memory");
    if (cell >= cellEnd || slot >= cellsPerXcd) return cpw;
    const int iniX = rec[1] & 0xFFFF, iniY = rec[1] >> 16;
    const int W = rec[2] & 0xFF, H = (rec[2] >> 8) & 0xFF, cellCap = rec[2] >> 17;
    int32_t* cellCnt = cellCountBase + ((uint64_t)frame * (uint32_t)totalCells + (rec[0] >> 4));     // the record names its cell
    if (rec[2] & 0x10000u) {                             // skipped cell (or a padding position of the balanced table)
        if (lane == 0 && (rec[0] >> 4) != 0x0FFFFFFFu) *cellCnt = 0;
        return cpw;
    }
    const int IW = W - 6, IH = H - 6;
    // window start (iniY, iniX - 1) of this frame; level 0 may be the caller's buffer
    const bool l0 = (rec[0] & 15u) == 0;
    const int pitch = l0 ? l0pitch : (int)rec[3];
    const uint8_t* win = l0 ? l0img + (int64_t)frame * l0fstride + (int64_t)iniY * l0pitch + (iniX - 1)
                            : pyrBase + (((uint64_t)rec[5] << 32) | rec[4]) + (uint64_t)frame * rec[6];

    // (the LDS base through an opaque zero per cell: as a loop invariant it and the offsets derived from it were kept in SGPRs
    //  across the cell body and spilled into VGPR lanes -- a v_readlane in front of every m0 write)
    int ldsZero = 0;
    asm volatile("" : "+s"(ldsZero));
    uint8_t* tile = pg_fast_smem + ldsZero + wv * waveLds;         // [tileRows][TP], this wave's slice
    uint8_t* smap = tile + tileRows * TP;                          // [mapRows][mapPitch], 1-px zero rim
    uint16_t* list = reinterpret_cast<uint16_t*>(smap + mapRows * mapPitch);   // [FAST_LIST_CAP]

    // (1) stage the window with LDS-DMA (global_load_lds, 16 B per lane): no VGPR round trip, no
    // alignment fix-up -- the per-lane GLOBAL address may be byte-unaligned (measured,
    // tools/ubench/glds_unaligned.hip), the LDS side is wave base + lane * 16, i.e. TP/16 lanes
    // per row and 64/(TP/16) rows per instruction.  LDS byte 0 of a row is global x = iniX - 1, so
    // dword j of a row holds window columns 4j-1 .. 4j+2 and interior quads are dword aligned.
    // Rows are read TP bytes wide: past the window that is the neighbouring cell / next row of the
    // level, always inside the level (the window ends >= 16 rows above the level's last row).
    {
        const int CH = TP >> 4, rowsPer = 64 / CH;
        const int r0 = (lane * chunkInv) >> 16, ch = lane - r0 * CH;       // lane / CH, lane % CH
        // the lane's offset from the window start fits 32 bits (a window is a few dozen rows): scalar 64-bit base +
        // 32-bit lane offset lets the load take its base from SGPRs (no 64-bit VALU address arithmetic per instruction)
        const uint32_t voff = (uint32_t)(r0 * pitch + ch * 16);
        const int nz = (mapRows * mapPitch + 15) >> 4;
        const bool laneOn = r0 < rowsPer;
        // Round 5: in the common geometry the first TWO loads (rows 0-20, 21-41) are issued by every lane with a chunk, whatever the
        // window's height: a row past the window is CLAMPED to its last row (a duplicate read of a line that is being fetched anyway), so
        // there is no guard per instruction -- three of them cost 36 scalar instructions and three exec round trips on a wave whose life
        // is its instruction count.  Rows 37-41 of a 37-row window land in the first bytes of the score map, which is therefore cleared
        // AFTER the landing now (below); a third load only exists for windows taller than 42 rows (wave-uniform branch).
        constexpr bool CLAMPED = TPC == 48 && MPC == 40 && NARROW;
        if (!CLAMPED) {
            // the score map is cleared FIRST (16 B per lane and step; the map starts 16-byte aligned and the candidate list
            // behind it absorbs the last partial step): behind the window loads the compiler waits for the DMA before every
            // ds_write (it cannot tell the two LDS targets apart), which put the clearing after the landing instead of under it
            for (int i = lane; i < nz; i += 64) reinterpret_cast<uint4*>(smap)[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (TPC == 48) {
            // 3 chunks per row, 21 rows per instruction, written out so that the address is SGPR base + 32-bit VGPR offset (the
            // builtin takes a flat 64-bit VGPR address).  A 36-px-wide cell of a level with ONE cell row can be up to 59 px tall (65
            // window rows): the wide instantiation unrolls six guarded steps (126 rows).  (Round 3's first version stopped at three for
            // both and lost the bottom rows of such cells -- found by the fuzz soak, tests/test_gpu_parity.py has the case now.)
            const uint32_t ldsTile = (uint32_t)(uintptr_t)(pg_lptr_t)tile;
            constexpr int KMAX = NARROW ? 3 : 6;
            if (CLAMPED) {
                const uint32_t chOff = (uint32_t)(ch * 16);
                const uint32_t v0 = (uint32_t)imad24(min(r0, H - 1), pitch, (int)chOff);          // (rows < 64, pitch < 2^23: one v_mad_i32_i24)
                const uint32_t v1 = (uint32_t)imad24(min(r0 + 21, H - 1), pitch, (int)chOff);
                if (laneOn) {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(v0), "s"(win), "s"(ldsTile) : "memory");
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(v1), "s"(win), "s"(ldsTile + 21 * 48) : "memory");
                }
            }
#pragma unroll
            for (int k = CLAMPED ? 2 : 0; k < KMAX; k++) {
                if (k * 21 < H) {                                          // wave-uniform
                    const uint8_t* gk = win + (int64_t)(k * 21) * pitch;   // scalar
                    int rk = r0;
                    if (CLAMPED) asm volatile("" : "+v"(rk));              // (the lane's guard is computed INSIDE the branch: windows taller than 42 rows are rare)
                    if (laneOn && rk + k * 21 < H)
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                     :: "v"(voff), "s"(gk), "s"(ldsTile + k * 21 * 48) : "memory");      // (m0 is reserved: the compiler never keeps a value in it across statements, and this instantiation has no other user)
                }
            }
        } else {
            for (int k = 0; k * rowsPer < H; k++) {
                const uint8_t* gk = win + (int64_t)(k * rowsPer) * pitch;
                if (laneOn && r0 + k * rowsPer < H)
                    __builtin_amdgcn_global_load_lds((pg_gptr_t)(gk + voff), (pg_lptr_t)(tile + k * rowsPer * TP), 16, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);                     // vmcnt(0): the DMA has landed
        if (CLAMPED) {                                     // <= 42 rows of 40 bytes: two unconditional steps of 1 KiB -- what they clear
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);    // past the map is the candidate list, written later (the launcher keeps
            reinterpret_cast<uint4*>(smap)[lane] = z;      // map + list >= 2 KiB)
            reinterpret_cast<uint4*>(smap)[lane + 64] = z;
        }
    }
    PG_WAVE_SYNC();
    uint32_t* out = cellCandBase + ((uint64_t)frame * cellCandFrame + rec[7]);
    const PgCellValid cv = {rec[8], rec[9], rec[10], rec[11], rec[12], rec[13], rec[14], rec[15]};
    const PgLaneValid valid = NARROW ? pg_lane_valid(cv) : PgLaneValid{0u, 0u};
    const int xoff = 3 + iniX - PG_EDGE, yoff = 3 + iniY - PG_EDGE;   // window-local -> region-relative (:822-823)

    const int total = fast_cell_detect<TPC, MPC, NARROW>(statusPtr, tile, TP, smap, mapPitch, mapRows, IW, IH, iniTh, minTh, list, out, cellCap,
                                                         xoff, yoff, lane, valid);
    if (lane == 0) *cellCnt = min(total, cellCap);
    return cpw;
    };      // one_cell
    // (the number of records comes out of the first cell's argument load: a load of its own in front would be a third scalar
    //  round trip on every wave)
    int j = 0, ncell;
#pragma unroll 1
    do {
        if (j) PG_WAVE_SYNC();
        ncell = one_cell(j);
    } while (++j < ncell);
}

void pg_launch_fast_cells(const PgPlan& P, int nframes, hipStream_t s, int levelBeg, int levelEnd);

// K2 for all levels of all frames (the block form of rounds 1-3 -- one workgroup per block of cells, 1.4 x slower on every
// shape, DESIGN.md section 6 -- left the tree in round 4)
void pg_launch_fast(const PgPlan& P, int nframes, hipStream_t s) { pg_launch_fast_cells(P, nframes, s, 0, P.nlevels); }

// K2 for the levels [levelBeg, levelEnd) only
void pg_launch_fast_levels(const PgPlan& P, int nframes, int levelBeg, int levelEnd, hipStream_t s)
{
    pg_launch_fast_cells(P, nframes, s, levelBeg, levelEnd);
}

void pg_launch_fast_cells(const PgPlan& P, int nframes, hipStream_t s, int levelBeg, int levelEnd)
{
    int maxW = 0, maxH = 0;
    for (int l = 0; l < P.nlevels; l++) {
        maxW = max(maxW, P.lvl[l].wCell + 6);
        maxH = max(maxH, P.lvl[l].hCell + 6);
    }
    int TP = (maxW + 6 + 15) & ~15;                        // byte 0 pad + window + quick-test over-read, 16-B chunks
    const int tileRows = maxH;
    int mapPitch = ((maxW - 6 + 2) + 3) & ~3;
    // tile-shape sweep (BASELINE.json configs[2]; pgorb_set_option "fast_tile_pitch"): a forced window pitch takes the
    // run-time-pitch instantiation -- 48 there measures what the immediate offsets are worth, 64 / 80 / ... what a wider LDS row costs
    const bool forced = P.fastTilePitch >= TP;
    if (forced) TP = P.fastTilePitch;
    const bool common = !forced && TP <= 48 && mapPitch <= 40 && maxH <= 126;        // the instantiation with immediate offsets (six window loads of 21 rows)
    if (common) { TP = 48; mapPitch = 40; }
    const int chunkInv = 65536 / (TP >> 4) + 1;            // lane / (TP/16) == (lane * chunkInv) >> 16 for lane < 64
    const int mapRows = maxH - 6 + 2;
    size_t smem = (size_t)tileRows * TP + max((size_t)mapRows * mapPitch + FAST_LIST_CAP * 2, (size_t)2048) + 16;      // (2 KiB: k_fast_cells clears the map in two whole steps)
    // profiling knob: extra LDS per wave lowers occupancy (occupancy sweep)
    if (const char* e = getenv("PGORB_FAST_EXTRA_LDS")) smem += (size_t)atoi(e);
    const bool all = levelBeg == 0 && levelEnd == P.nlevels && P.cellTabBal;
    const uint32_t* tab = all ? P.cellTabBal : P.cellTab;
    const int cell0 = all ? 0 : P.lvl[levelBeg].cellBase;
    const int cellEnd = all ? 8 * P.cellsPerXcdBal : (levelEnd < P.nlevels) ? P.lvl[levelEnd].cellBase : P.totalCells;
    const int cellsPerXcd = all ? P.cellsPerXcdBal : (cellEnd - cell0 + 7) / 8;
    const bool narrow = maxW - 6 <= 32 && maxH - 6 <= 40;      // 8 quads per row, at most 5 steps of 8 rows (quick_pass_b)
    int wpb = (P.fastWpb == 4) ? 4 : (P.fastWpb == 2) ? 2 : 1; // 4 independent waves per workgroup measured 13 % slower (round 2), 2: round 5
    if (const char* e = getenv("PGORB_FAST_WPB")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) wpb = v; }
    const int waveLds = (int)((smem + 15) & ~(size_t)15);
    int cpw = P.fastCpw > 0 ? P.fastCpw : 1;                   // records per wave (option "fast_cells_per_wave"; PGORB_FAST_CPW overrides)
    if (const char* e = getenv("PGORB_FAST_CPW")) cpw = atoi(e);
    cpw = cpw < 1 ? 1 : (cpw > 64 ? 64 : cpw);
    dim3 grid(((cellsPerXcd + wpb * cpw - 1) / (wpb * cpw)) * 8, nframes), block(64 * wpb);
    PgFastArgs KA;
    memset(&KA, 0, sizeof KA);
    KA.l0img = P.lvl[0].img; KA.l0fstride = P.lvl[0].fstride; KA.pyrBase = P.pyrBase; KA.cellCount = P.cellCount;
    KA.cellCand = P.cellCand; KA.status = P.status; KA.tab = tab; KA.l0pitch = P.lvl[0].pitch; KA.totalCells = P.totalCells;
    KA.cellCandFrame = (uint32_t)P.cellCandFrame; KA.iniTh = P.iniTh; KA.minTh = P.minTh; KA.TPr = TP; KA.tileRows = tileRows;
    KA.MPr = mapPitch; KA.mapRows = mapRows; KA.cellsPerXcd = cellsPerXcd; KA.chunkInv = chunkInv; KA.cell0 = cell0; KA.cellEnd = cellEnd;
    KA.waveLds = waveLds; KA.cpw = cpw;
#define PG_LAUNCH_CELLS(TPC, MPC, NAR, W) hipLaunchKernelGGL((k_fast_cells<TPC, MPC, NAR, W>), grid, block, (size_t)waveLds * W, s, KA)
    if (wpb == 2) {
        if (common && narrow) PG_LAUNCH_CELLS(48, 40, true, 2);
        else if (common) PG_LAUNCH_CELLS(48, 40, false, 2);
        else PG_LAUNCH_CELLS(0, 0, false, 2);
    } else if (wpb == 4) {
        if (common && narrow) PG_LAUNCH_CELLS(48, 40, true, 4);
        else if (common) PG_LAUNCH_CELLS(48, 40, false, 4);
        else PG_LAUNCH_CELLS(0, 0, false, 4);
    } else {
        if (common && narrow) PG_LAUNCH_CELLS(48, 40, true, 1);
        else if (common) PG_LAUNCH_CELLS(48, 40, false, 1);
        else PG_LAUNCH_CELLS(0, 0, false, 1);
    }
#undef PG_LAUNCH_CELLS
}

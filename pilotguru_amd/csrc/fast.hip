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
//      the wave finds its window through one 32-byte cell record (a single scalar load);
//      (persistent waves that prefetch the next window measured slower twice -- with register
//      prefetch 2x, with LDS-DMA double buffering 1.5x: the kernel is VALU-issue bound and
//      needs its 8 waves per SIMD more than it needs the load latency hidden)
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

extern __shared__ __attribute__((aligned(16))) uint8_t pg_fast_smem[];
typedef __attribute__((address_space(1))) const void* pg_gptr_t;
typedef __attribute__((address_space(3))) void* pg_lptr_t;

// v_min3_i32 / v_max3_i32 as opaque instructions: written as min(min(a, b), c) the optimiser re-associates the
// sliding-window chains of the exact score, shares two-input pairs between neighbouring windows and ends up with MORE
// instructions (63 three-input + 26 two-input per score instead of 80 three-input), all in the slow issue class
__device__ __forceinline__ int imin3(int a, int b, int c)
{
    int r;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int imax3(int a, int b, int c)
{
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int imad24(int a, int b, int c)        // a * b + c on 24-bit operands: one instruction
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 16-ring offsets in OpenCV's order (x, y): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)
// (0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
__device__ __forceinline__ void ring_load(const uint8_t* c, int p, int v, int d[16])
{
    d[0] = v - c[3 * p];          d[1] = v - c[3 * p + 1];    d[2] = v - c[2 * p + 2];
    d[3] = v - c[p + 3];          d[4] = v - c[3];            d[5] = v - c[-p + 3];
    d[6] = v - c[-2 * p + 2];     d[7] = v - c[-3 * p + 1];   d[8] = v - c[-3 * p];
    d[9] = v - c[-3 * p - 1];     d[10] = v - c[-2 * p - 2];  d[11] = v - c[-p - 3];
    d[12] = v - c[-3];            d[13] = v - c[p - 3];       d[14] = v - c[2 * p - 2];
    d[15] = v - c[3 * p - 1];
}

// OpenCV cornerScore<16> for a corner: max over the 16 nine-long arcs of the arc minimum of
// d (darker ring) or of -d (brighter ring), minus 1.
__device__ __forceinline__ int fast_score16(const int d[16])
{
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = imin3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        hi3[k] = imax3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    }
    // 9-window minima / maxima, then a 3-input reduction tree: 32 + 32 + 16 three-input operations in all
    // (v_min3 / v_max3 issue in the slow class like the two-input forms, so every fused pair is a slot saved)
    int lo9[16], hi9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo9[k] = imin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
        hi9[k] = imax3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
    }
    int best_dark = imax3(lo9[0], lo9[1], lo9[2]), best_bright = imin3(hi9[0], hi9[1], hi9[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) {
        best_dark = imax3(best_dark, lo9[k], lo9[k + 1]);
        best_bright = imin3(best_bright, hi9[k], hi9[k + 1]);
    }
    best_dark = max(best_dark, lo9[15]);
    best_bright = min(best_bright, hi9[15]);
    return max(best_dark, -best_bright) - 1;
}

// inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (6 cross-lane moves, no
// LDS crossbar): Hillis-Steele inside each row of 16, then row totals carried with
// row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3)
__device__ __forceinline__ int wave_incl_scan(int x)
{
    int v = x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31
    return v;
}

__device__ __forceinline__ int wave_prefix(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// wave-local "barrier": the cell form never synchronises ACROSS waves (one wave = one cell); LDS operations of one
// wave are executed in order, so all it needs is that the compiler does not move LDS accesses across the point
#define PG_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// tile layout: row r at tile + r*TP; window column c at byte 1 + c, so interior column 0
// (window column 3) is at byte 4.
#define FAST_LIST_CAP 768          // compacted candidates held in LDS (u16 each)
#ifndef PG_FAST_COMPACT_BALLOT      // developer A/B (tools/experiments/r5_k2_ab.sh): 1 = compaction by per-trip ballots, 0 = DPP prefix scan + per-lane runs
#define PG_FAST_COMPACT_BALLOT 1
#endif

typedef unsigned short pg_us2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pg_pkmin(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(pg_us2f, a), __builtin_bit_cast(pg_us2f, b)));
}
__device__ __forceinline__ uint32_t pg_pkmax(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pg_us2f, a), __builtin_bit_cast(pg_us2f, b)));
}

// (2) necessary test + compaction of interior rows [rowBeg, rowEnd).  Each lane tests one quad
// per step.  A pixel can only be a corner at threshold t if both opposite ring pairs (0,8) and
// (4,12) hold a pixel darker than v-t, or both a pixel brighter than v+t.  The test runs on
// TWO pixels per 32-bit operation: v_perm_b32 spreads the bytes of a dword into 16-bit fields
// (pixels 0,2 / 1,3, and the ring pixels x-3 / x+3 straight from the neighbouring dwords),
// v_pk_min_u16 / v_pk_max_u16 form dk = the larger of the pairs' minima and br = the smaller of
// their maxima, and for a field
//     0x8000 + v - t - 1 - dk  has bit 15 set  <=>  dk < v - t       (a darker pixel in every pair)
//     0x8000 + br - v - t - 1  has bit 15 set  <=>  br > v + t       (a brighter pixel in every pair)
// with no borrow or carry between fields (|v - r| + t + 1 < 0x8000).
// The 4 result bits of a step are kept in a 64-bit register (step s: bits 14-2s, 15-2s, 30-2s,
// 31-2s of word s/8 = pixels 0,1,2,3); ONE wave prefix sum at the end turns the per-lane
// popcounts into list offsets (list order is irrelevant: NMS works on the score map).  Returns
// the list length, or -1 when the list would overflow (the caller then takes the chunked path).
// STRONG adds the diagonal pairs (2,10) and (6,14) to the necessary condition (a 9-long arc holds
// one pixel of EVERY opposite pair).  It costs ~35 more operations per step and is used for the
// minThFAST pass, where the two-pair test lets ~40 % of a textured cell through and the exact
// scores of those pixels dominated the pass.
template <int QW, bool STRONG>      // quads per row handled by consecutive lanes: 8 (IW <= 32) or 16 (IW <= 64)
__device__ __forceinline__ int quick_pass(const uint8_t* tile, int TP, int IW, int rowBeg, int rowEnd,
                                          int t, uint16_t* list, int lane)
{
    // lane -> (quad, row): with 8 quads per row the 32 lanes of a bank group (ds_read_b32: lanes 0-31 / 32-63) take
    // rows 0, 2, 4, 6 (resp. 1, 3, 5, 7) of the step: at the 48-byte tile pitch (12 dwords) those start on banks
    // 0, 24, 16, 8 -- four disjoint runs of 8 banks.  Consecutive rows (0, 12, 24, 36 -> 4) put rows 0 and 3 on the
    // same banks (SQ_LDS_BANK_CONFLICT was 44 % of SQ_LDS_IDX_ACTIVE, profiles/r01_i_pmc.txt).
    const int lq = lane & (QW - 1);
    const int lr = (QW == 8) ? (((lane >> 3) & 3) * 2 + (lane >> 5)) : lane / QW;
    const int NQ = (IW + 3) >> 2;
    const uint32_t K15 = 0x80008000u;
    const uint32_t Kd = (uint32_t)(0x8000 - t - 1) * 0x00010001u;
    // columns of this quad inside the interior, in result-bit layout
    uint32_t colMask = 0;
    if (4 * lq + 0 < IW) colMask |= 1u << 14;
    if (4 * lq + 1 < IW) colMask |= 1u << 15;
    if (4 * lq + 2 < IW) colMask |= 1u << 30;
    if (4 * lq + 3 < IW) colMask |= 1u << 31;
    // the darker-ring and the brighter-ring outcome are kept apart: a list entry carries the polarity that passed
    // (bit 15), and the exact score then evaluates ONE polarity -- 40 instead of 80 three-input operations.  A pixel
    // that passes both ways (rare) gets two entries; at most one of them can reach the threshold.
    uint32_t accD[2] = {0u, 0u}, accB[2] = {0u, 0u};
    int step = 0;
    for (int row0 = rowBeg; row0 < rowEnd; row0 += 64 / QW, step++) {
        const int iy = row0 + lr;
        uint32_t md = 0, mb = 0;
        if (iy < rowEnd && lq < NQ) {
            const uint32_t* rc = reinterpret_cast<const uint32_t*>(tile + (iy + 3) * TP) + 1 + lq;
            const uint32_t* ru = reinterpret_cast<const uint32_t*>(tile + iy * TP) + 1 + lq;
            const uint32_t* rd = reinterpret_cast<const uint32_t*>(tile + (iy + 6) * TP) + 1 + lq;
            const uint32_t C = rc[0], Lw = rc[-1], Rw = rc[1], U = ru[0], D = rd[0];
            // one v_perm_b32 per operand: bytes -> two 16-bit fields (selector 0x0c = zero byte)
            const uint32_t OD = 0x0c030c01u;                          // (b1, b3) of one dword
            const uint32_t X20 = 0x0c040c02u, X31 = 0x0c050c03u;     // (lo.b2, hi.b0) / (lo.b3, hi.b1) of a dword pair
            // (the even-byte fields are one v_and_b32 -- gfx950 issues and / or / add / sub / lshr at 2.7 cycles per wave
            //  instruction, v_perm_b32 and everything packed or 3-operand at 4.6: tools/ubench/valu_rate3.hip)
            const uint32_t M02 = 0x00FF00FFu;
            const uint32_t Ce = C & M02, Co = __builtin_amdgcn_perm(C, C, OD);   // pixels (0,2) / (1,3)
            const uint32_t Ue = U & M02, Uo = __builtin_amdgcn_perm(U, U, OD);   // ring 8 (3 rows up)
            const uint32_t De = D & M02, Do = __builtin_amdgcn_perm(D, D, OD);   // ring 0 (3 rows down)
            // ring 12 (x-3) and ring 4 (x+3) of the even and odd pixels
            const uint32_t W12e = __builtin_amdgcn_perm(Lw, Lw, OD);   // pixels (-3, -1)
            const uint32_t W4o = Rw & M02;                             // pixels (4, 6)
            const uint32_t W12o = __builtin_amdgcn_perm(C, Lw, X20);   // pixels (-2, 0)
            const uint32_t W4e = __builtin_amdgcn_perm(Rw, C, X31);    // pixels (3, 5)
            // per parity: dk = the larger of the pairs' minima, br = the smaller of their maxima
            // (v_pk_min_u16 / v_pk_max_u16 on the 16-bit fields); darker-in-every-pair <=> dk < v - t,
            // brighter-in-every-pair <=> br > v + t, each decided by bit 15 of one add / sub
            uint32_t dkE = pg_pkmax(pg_pkmin(De, Ue), pg_pkmin(W4e, W12e)), brE = pg_pkmin(pg_pkmax(De, Ue), pg_pkmax(W4e, W12e));
            uint32_t dkO = pg_pkmax(pg_pkmin(Do, Uo), pg_pkmin(W4o, W12o)), brO = pg_pkmin(pg_pkmax(Do, Uo), pg_pkmax(W4o, W12o));
            if (STRONG) {
                // rows y+2 / y-2 at x+2 / x-2: rings 2 (+2,+2), 14 (-2,+2), 6 (+2,-2), 10 (-2,-2)
                const uint32_t* rp = reinterpret_cast<const uint32_t*>(tile + (iy + 5) * TP) + 1 + lq;
                const uint32_t* rm = reinterpret_cast<const uint32_t*>(tile + (iy + 1) * TP) + 1 + lq;
                const uint32_t Pc = rp[0], Pl = rp[-1], Pr = rp[1], Mc = rm[0], Ml = rm[-1], Mr = rm[1];
                const uint32_t r2e = __builtin_amdgcn_perm(Pr, Pc, X20), r14e = __builtin_amdgcn_perm(Pc, Pl, X20);   // x+2 / x-2 of pixels (0,2)
                const uint32_t r6e = __builtin_amdgcn_perm(Mr, Mc, X20), r10e = __builtin_amdgcn_perm(Mc, Ml, X20);
                const uint32_t r2o = __builtin_amdgcn_perm(Pr, Pc, X31), r14o = __builtin_amdgcn_perm(Pc, Pl, X31);   // ... of pixels (1,3)
                const uint32_t r6o = __builtin_amdgcn_perm(Mr, Mc, X31), r10o = __builtin_amdgcn_perm(Mc, Ml, X31);
                dkE = pg_pkmax(dkE, pg_pkmax(pg_pkmin(r2e, r10e), pg_pkmin(r6e, r14e)));
                brE = pg_pkmin(brE, pg_pkmin(pg_pkmax(r2e, r10e), pg_pkmax(r6e, r14e)));
                dkO = pg_pkmax(dkO, pg_pkmax(pg_pkmin(r2o, r10o), pg_pkmin(r6o, r14o)));
                brO = pg_pkmin(brO, pg_pkmin(pg_pkmax(r2o, r10o), pg_pkmax(r6o, r14o)));
            }
            const uint32_t darkE = (Ce + Kd) - dkE, brightE = brE + (Kd - Ce);
            const uint32_t darkO = (Co + Kd) - dkO, brightO = brO + (Kd - Co);
            md = (((darkE & K15) >> 1) | (darkO & K15)) & colMask;
            mb = (((brightE & K15) >> 1) | (brightO & K15)) & colMask;
        }
        if (QW == 8) { accD[0] |= md >> (2 * step); accB[0] |= mb >> (2 * step); }      // <= 48 rows = 6 steps: one word
        else { accD[step >> 3] |= md >> (2 * (step & 7)); accB[step >> 3] |= mb >> (2 * (step & 7)); }
    }
    const int cnt = __popc(accD[0]) + __popc(accD[1]) + __popc(accB[0]) + __popc(accB[1]);
    const int incl = wave_incl_scan(cnt);
    const int nlist = __builtin_amdgcn_readlane(incl, 63);
    if (nlist > FAST_LIST_CAP) return -1;
    int off = incl - cnt;
#pragma unroll
    for (int pol = 0; pol < 2; pol++)
#pragma unroll
        for (int wsel = 0; wsel < 2; wsel++) {
            uint32_t bits = pol ? accB[wsel] : accD[wsel];
            while (bits) {
                const int bpos = __ffs((int)bits) - 1;
                bits &= bits - 1;
                const int st = wsel * 8 + 7 - ((bpos & 15) >> 1);
                const int iy = rowBeg + st * (64 / QW) + lr;                                      // < 128
                list[off++] = (uint16_t)((pol << 15) | (iy << 8) | (4 * lq + ((bpos >> 4) << 1) + (bpos & 1)));
            }
        }
    return nlist;
}

// (2b) round 3: the necessary test of the narrow geometry on FOUR pixels per 32-bit operation, bytes in place -- no
// unpacking into 16-bit fields at all.  With nC = ~C (the quad's four centre pixels, complemented) one v_lerp_u8 per ring
// operand R (the dword of the four ring pixels that belong to the four centres) gives, byte for byte, without carries
//     E = (R + 255 - C + 1) >> 1 = 128 + floor((R - C) / 2)
// and with T = t + 1:  ring darker   (R - C <= -T)  =>  E <= 128 + floor(-T / 2) = Kd   (exact for odd T, one grey level
//                                                                                         weaker for even T)
//                      ring brighter (R - C >=  T)  =>  E >= 128 + floor( T / 2) = Kb   (exact for even T, one weaker for odd)
// -- a NECESSARY condition is all this stage has to be (the exact score decides), the weaker side lets ~4 % more
// pixels through.  The two byte-wise threshold compares run on the low seven bits, where an add cannot carry into the
// next byte:   low = E & 0x7f..;  A = low + (256 - Kb): bit 7 <=> low >= Kb - 128;  Q = low + (127 - Kd): bit 7 <=> low > Kd
//              brighter <=> bit 7 of (E & A),   darker <=> bit 7 of ~(E | Q).
// Ring operands: rows y-3 / y+3 are the dwords above / below (no shuffle), x-3 / x+3 are one v_alignbyte_b32 each of the
// centre row's three dwords; STRONG adds the diagonals (x+-2, y+-2), four more v_alignbyte_b32.  Per step: 6 slow-class
// (lerp, alignbyte) + 28 fast-class (and / add / bitop3 / shift) operations against 18 + 18 in the 16-bit-field form
// (tools/ubench/valu_rate4.hip: v_lerp_u8, v_alignbyte_b32, v_perm_b32, v_pk_* 4.7 cycles, the others 2.7).
// Result bits: byte p of the accumulator = pixel p of the quad, bit 4 + s = darker-ring outcome of step s, bit s =
// brighter-ring outcome (s = 0..3; a fifth step, interiors taller than 32 rows, fills a second word).  ONE 32-bit word
// holds a lane's 16 pixels x 2 polarities, so the compaction loop runs max-over-lanes(count) times instead of once per
// polarity, and a list entry is just (a per-lane constant) | bit position: the (row, column) arithmetic is done
// by the scoring round, for 64 entries at once, not per entry by the lane that found it.
// entry = fifth-step flag (13) | row of the lane (8-10) | quad of the lane (5-7) | bit position (0-4: pixel, polarity, step):
// the column 4 * quad + pixel is ONE bit field of the entry, and entry >> 8 = lane row + 32 * (fifth step) is the row but for
// the 8 * step of bits 0-1 -- three operations to decode a row
#define BFMT_FIFTH 0x2000
__device__ __forceinline__ int bfmt_lane_base(int lane)
{
    const int lr = ((lane >> 3) & 3) * 2 + (lane >> 5);              // the lane -> row map of the test (see quick_pass)
    return (lr << 8) | ((lane & 7) << 5);
}
__device__ __forceinline__ void bfmt_decode(int e, int& iy, int& ix, int& bright)
{
    ix = (e >> 3) & 31;                                              // 4 * quad + pixel
    iy = ((e & 3) << 3) | (e >> 8);                                  // 8 * step + lane row (+ 32: fifth step)
    bright = ((e >> 2) & 1) ^ 1;
}

// the cell record's validity words (PgPlan::cellTab w8-w15), as the scalars they are
struct PgCellValid { uint32_t qLt, qEq, partial, rowLo, rowHi, stepBase, stepMore, fifth; };
// 0 / ~0 per lane from a 64-bit LANE mask held in SGPRs: one v_cndmask_b32 (the mask is the instruction's condition operand)
__device__ __forceinline__ uint32_t pg_lanes(unsigned long long m)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(r) : "s"(m));
    return r;
}

// ... turned into this lane's words once per cell (both detector passes use them): pixels of the lane's quad inside the interior (whole
// bytes) x steps whose row 8 s + lr lies inside it.  10 vector instructions instead of the 17 that derived them from IW, IH, lq, lr
struct PgLaneValid { uint32_t acc, acc2; };
__device__ __forceinline__ PgLaneValid pg_lane_valid(const PgCellValid& V)
{
    const unsigned long long mRow = ((unsigned long long)V.rowHi << 32) | V.rowLo;
    const uint32_t colBytes = (pg_lanes(((unsigned long long)V.qEq << 32) | V.qEq) & V.partial) | pg_lanes(((unsigned long long)V.qLt << 32) | V.qLt);
    const uint32_t stepBits = (pg_lanes(mRow) & V.stepMore) | V.stepBase;
    PgLaneValid L;
    L.acc = colBytes & stepBits;
    L.acc2 = colBytes & pg_lanes(V.fifth == 2 ? ~0ull : (V.fifth == 1 ? mRow : 0ull));
    return L;
}

template <bool STRONG>
__device__ __forceinline__ int quick_pass_b(const uint8_t* tile, int IW, int IH, int t, uint16_t* list, int lane, const PgLaneValid& V)
{
    constexpr int TP = 48;
    const int lq = lane & 7;
    const int lr = ((lane >> 3) & 3) * 2 + (lane >> 5);
    const int T = t + 1;
    const uint32_t M7 = 0x7F7F7F7Fu, M80 = 0x80808080u, ONES = 0x01010101u;
    const uint32_t KA = (uint32_t)(256 - (128 + (T >> 1))) * ONES;          // 256 - Kb
    const uint32_t KQ = (uint32_t)(127 - (128 - ((T + 1) >> 1))) * ONES;    // 127 - Kd,  Kd = 128 - ceil(T / 2)
    const uint32_t* b0 = reinterpret_cast<const uint32_t*>(tile + lr * TP) + 1 + lq;   // (row lr, interior quad lq)
    uint32_t acc = 0u, acc2 = 0u;
#define PG_RING(R, E, A, Q) const uint32_t E = __builtin_amdgcn_lerp(R, nC, ONES); \
                            const uint32_t A = (E & M7) + KA, Q = (E & M7) + KQ
    // the eleven dwords a step reads (five for the two-pair form); the retry's four-pair form loads them ONE STEP AHEAD (below)
    struct Rows { uint32_t C, Lw, Rw, U, D, Pc, Pl, Pr, Mc, Ml, Mr; };
    auto load_rows = [&](const int s) {
        Rows r;
        const uint32_t* ru = b0 + (8 * s) * (TP / 4);
        const uint32_t* rc = b0 + (8 * s + 3) * (TP / 4);
        const uint32_t* rd = b0 + (8 * s + 6) * (TP / 4);
        r.C = rc[0]; r.Lw = rc[-1]; r.Rw = rc[1]; r.U = ru[0]; r.D = rd[0];
        if (STRONG) {
            // diagonals: rows y+2 / y-2 at x+2 / x-2 -- rings 2 (+2,+2), 14 (-2,+2), 6 (+2,-2), 10 (-2,-2); opposite pairs (2,10), (6,14)
            const uint32_t* rp = b0 + (8 * s + 5) * (TP / 4);
            const uint32_t* rm = b0 + (8 * s + 1) * (TP / 4);
            r.Pc = rp[0]; r.Pl = rp[-1]; r.Pr = rp[1]; r.Mc = rm[0]; r.Ml = rm[-1]; r.Mr = rm[1];
        } else { r.Pc = r.Pl = r.Pr = r.Mc = r.Ml = r.Mr = 0u; }
        return r;
    };
    constexpr bool SPF = STRONG;
    Rows nxt = SPF ? load_rows(0) : Rows{};
#pragma unroll
    for (int s = 0; s < 5; s++) {
        // steps 0-3 run whatever the interior's height (their rows lie inside the LDS allocation: row 8 s + lr + 6 <= 37; what they
        // see past the interior is masked by V): no scalar branch between the steps, so the four steps are ONE basic block and
        // their LDS reads can all be in flight before the first lerp.  Only the fifth step (interiors taller than 32 rows) is optional.
        // (The four-pair form of the retry keeps its steps apart -- 44 reads in flight at once do not fit the 64 registers, and a
        //  sched_barrier between step pairs did not hold the reads back -- and has the NEXT step's eleven reads in flight while it
        //  computes this one's: one exposed LDS round trip per cell instead of four.)
        if (STRONG ? 8 * s >= IH : (s == 4 && IH <= 32)) break;       // wave-uniform
        const Rows R = SPF ? nxt : load_rows(s);
        if (SPF && s < 3) nxt = load_rows(s + 1);                     // (rows of steps <= 3 are always inside the allocation)
        else if (SPF && s == 3 && IH > 32) nxt = load_rows(4);        // wave-uniform
        if (SPF) PG_WAVE_SYNC();                                   // (compiler-only: keeps the reads HERE -- left alone they are sunk to their uses in the next step's block)
        const uint32_t C = R.C, Lw = R.Lw, Rw = R.Rw, U = R.U, D = R.D;
        const uint32_t nC = ~C;
        const uint32_t W12 = __builtin_amdgcn_alignbyte(C, Lw, 1);    // pixels x-3 of the quad: (Lw.b1, Lw.b2, Lw.b3, C.b0)
        const uint32_t W4 = __builtin_amdgcn_alignbyte(Rw, C, 3);     // pixels x+3: (C.b3, Rw.b0, Rw.b1, Rw.b2)
        PG_RING(U, eU, aU, qU); PG_RING(D, eD, aD, qD); PG_RING(W12, eL, aL, qL); PG_RING(W4, eR, aR, qR);
        // brighter in a pair: (E & A) of either member; in every pair: the AND of the pairs.     bitop3 0xF8 = a | (b & c)
        uint32_t br = __builtin_amdgcn_bitop3_b32(eU & aU, eD, aD, 0xF8) & __builtin_amdgcn_bitop3_b32(eL & aL, eR, aR, 0xF8);
        // NOT darker in a pair: (E | Q) of both members; darker in every pair <=> no pair's bit set.   0xE0 = a & (b | c)
        uint32_t nd = __builtin_amdgcn_bitop3_b32(eU | qU, eD, qD, 0xE0) | __builtin_amdgcn_bitop3_b32(eL | qL, eR, qR, 0xE0);
        if (STRONG) {
            const uint32_t Pc = R.Pc, Pl = R.Pl, Pr = R.Pr, Mc = R.Mc, Ml = R.Ml, Mr = R.Mr;
            const uint32_t r2 = __builtin_amdgcn_alignbyte(Pr, Pc, 2), r14 = __builtin_amdgcn_alignbyte(Pc, Pl, 2);
            const uint32_t r6 = __builtin_amdgcn_alignbyte(Mr, Mc, 2), r10 = __builtin_amdgcn_alignbyte(Mc, Ml, 2);
            PG_RING(r2, e2, a2, q2); PG_RING(r10, e10, a10, q10); PG_RING(r6, e6, a6, q6); PG_RING(r14, e14, a14, q14);
            br &= __builtin_amdgcn_bitop3_b32(e2 & a2, e10, a10, 0xF8) & __builtin_amdgcn_bitop3_b32(e6 & a6, e14, a14, 0xF8);
            nd |= __builtin_amdgcn_bitop3_b32(e2 | q2, e10, q10, 0xE0) | __builtin_amdgcn_bitop3_b32(e6 | q6, e14, q14, 0xE0);
        }
        // bit 7 := darker (= ~nd), bit 3 := brighter;  0x4E = c ? ~a : b with c = 0x80808080
        const uint32_t comb = __builtin_amdgcn_bitop3_b32(nd, br >> 4, M80, 0x4E);
        if (s < 4) acc = __builtin_amdgcn_bitop3_b32(acc, comb >> (3 - s), 0x88888888u >> (3 - s), 0xF8);
        else acc2 = (comb >> 3) & 0x11111111u;
    }
#undef PG_RING
    acc &= V.acc;                                                     // validity (pg_lane_valid)
    acc2 &= V.acc2;
#if defined(PGORB_FAST_STOP) && PGORB_FAST_STOP == 2
    return (__ballot((acc | acc2) != 0) != 0ull) ? 1 : 0;
#endif
#if PG_FAST_COMPACT_BALLOT
    // Compaction by ballots, one per trip (round 5): trip k files the k-th hit of every lane that has one -- position = hits filed so
    // far (a scalar) + the lane's rank among the lanes of this trip (v_mbcnt of the ballot).  No prefix scan in front (six dependent
    // DPP steps and their wait states), no popcounts; the list comes out trip-major, so a lane's own hits -- rows 8 apart at the
    // 48-byte pitch: the same LDS bank -- no longer sit next to each other in a score round.
    const int base = bfmt_lane_base(lane);
    int nlist = 0;                                                    // wave-uniform
    uint32_t bits = acc;
    for (;;) {
        const unsigned long long m = __ballot(bits != 0);
        if (!m) break;
        if (bits) {
            const int bpos = __ffs((int)bits) - 1;
            bits &= bits - 1;
            const int pos = nlist + wave_prefix(m);
            if (pos < FAST_LIST_CAP) list[pos] = (uint16_t)(base | bpos);
        }
        nlist += __popcll(m);
    }
    if (IH > 32) {                                                    // wave-uniform
        bits = acc2;
        for (;;) {
            const unsigned long long m = __ballot(bits != 0);
            if (!m) break;
            if (bits) {
                const int bpos = __ffs((int)bits) - 1;
                bits &= bits - 1;
                const int pos = nlist + wave_prefix(m);
                if (pos < FAST_LIST_CAP) list[pos] = (uint16_t)(base | BFMT_FIFTH | bpos);
            }
            nlist += __popcll(m);
        }
    }
    return nlist > FAST_LIST_CAP ? -1 : nlist;
#else
    const int cnt = __popc(acc) + __popc(acc2);
    const int incl = wave_incl_scan(cnt);
    const int nlist = __builtin_amdgcn_readlane(incl, 63);
    if (nlist > FAST_LIST_CAP) return -1;
    uint16_t* lp = list + (incl - cnt);
    const int base = bfmt_lane_base(lane);
    uint32_t bits = acc;
    while (bits) {
        const int bpos = __ffs((int)bits) - 1;
        bits &= bits - 1;
        *lp++ = (uint16_t)(base | bpos);
    }
    if (IH > 32) {                                                    // wave-uniform
        bits = acc2;
        while (bits) {
            const int bpos = __ffs((int)bits) - 1;
            bits &= bits - 1;
            *lp++ = (uint16_t)(base | BFMT_FIFTH | bpos);
        }
    }
    return nlist;
#endif
}

// (3) exact scores for the compacted pixels -> score map.  An entry names the polarity its pixel passed the necessary
// test with; with sgn = +1 (darker ring) / -1 (brighter ring) the differences d = sgn * (v - ring) make both cases the
// "darker" case: score = (largest 9-arc minimum of d) - 1, i.e. 16 v_mad_i32_i24 + 32 v_min3 + 8 v_max3 where both
// polarities cost 16 + 80.  Entries whose pixel is not a corner at t are overwritten with 0xFFFF: NMS skips them
// without touching the score map, and of a pixel's two entries (both polarities passed) at most one survives.
#define FAST_DEAD 0xFFFFu
#ifndef PG_FAST_NO_FUSED           // developer A/B builds of two round-5 changes (tools/experiments/r5_k2_ab.sh)
#define PG_FAST_NO_FUSED 0
#endif
#ifndef PG_FAST_NO_CLAMP
#define PG_FAST_NO_CLAMP 0
#endif
#ifndef PG_FAST_SCORE_I32          // developer A/B build (make EXTRA=-DPG_FAST_SCORE_I32=1): the 32-bit integer form of rounds 2-4 everywhere
#define PG_FAST_SCORE_I32 0
#endif
template <bool BFMT>     // entry format: quick_pass_b's (lane, bit position) or the older (polarity << 15) | (iy << 8) | ix
__device__ __forceinline__ void score_list(const uint8_t* tile, int TP, uint8_t* smap, int mapPitch,
                                           uint16_t* list, int nlist, int t, int lane)
{
    for (int base = 0; base < nlist; base += 64) {
        const int i = base + lane;
        if (i < nlist) {
            const int e = list[i];
            int iy, ix, bright;
            if (BFMT) bfmt_decode(e, iy, ix, bright);
            else { iy = (e >> 8) & 0x7F; ix = e & 0xFF; bright = e >> 15; }
            const int nsg = bright ? 1 : -1;                          // -sgn
            const uint8_t* c = tile + (iy + 3) * TP + 4 + ix;
            const int p = TP;
            const int sv = -nsg * (int)c[0];                          // sgn * v
            int d[16];
            // ring offsets in OpenCV's order (see ring_load); d = sgn * v - sgn * ring
            d[0] = imad24((int)c[3 * p], nsg, sv);        d[1] = imad24((int)c[3 * p + 1], nsg, sv);
            d[2] = imad24((int)c[2 * p + 2], nsg, sv);    d[3] = imad24((int)c[p + 3], nsg, sv);
            d[4] = imad24((int)c[3], nsg, sv);            d[5] = imad24((int)c[-p + 3], nsg, sv);
            d[6] = imad24((int)c[-2 * p + 2], nsg, sv);   d[7] = imad24((int)c[-3 * p + 1], nsg, sv);
            d[8] = imad24((int)c[-3 * p], nsg, sv);       d[9] = imad24((int)c[-3 * p - 1], nsg, sv);
            d[10] = imad24((int)c[-2 * p - 2], nsg, sv);  d[11] = imad24((int)c[-p - 3], nsg, sv);
            d[12] = imad24((int)c[-3], nsg, sv);          d[13] = imad24((int)c[p - 3], nsg, sv);
            d[14] = imad24((int)c[2 * p - 2], nsg, sv);   d[15] = imad24((int)c[3 * p - 1], nsg, sv);
            int lo3[16], lo9[16];
#pragma unroll
            for (int k = 0; k < 16; k++) lo3[k] = imin3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
            for (int k = 0; k < 16; k++) lo9[k] = imin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
            int best = imax3(lo9[0], lo9[1], lo9[2]);
#pragma unroll
            for (int k = 3; k < 15; k += 2) best = imax3(best, lo9[k], lo9[k + 1]);
            const int s = max(best, lo9[15]) - 1;
            const bool corner = s >= t;
            if (corner) smap[(iy + 1) * mapPitch + ix + 1] = (uint8_t)s;
            // NMS reads (iy << 8) | ix: corners are re-filed in that form, everything else is marked dead
            if (BFMT) list[i] = corner ? (uint16_t)((iy << 8) | ix) : (uint16_t)FAST_DEAD;
            else if (!corner) list[i] = (uint16_t)FAST_DEAD;
        }
    }
}

// (3') round 5: the exact score on PACKED f16 pairs -- 21 three-input operations per candidate instead of 16 + 40.
// A FAST score only ORDERS ring pixels, so the differences never have to be formed: with x_k = sgn * (1024 + ring_k) (f16: the
// bit pattern 0x6400 | byte IS 1024 + byte, the sign bit negates -- one v_bitop3_b32 per register, no conversion)
//     max over the 16 arcs of (min over the arc of sgn * (v - ring))  =  sgn * (1024 + v)  -  min over the arcs of (max over the arc of x),
// for the darker ring (sgn = +1) and the brighter ring (sgn = -1) alike.  Ring positions k and k + 8 share a register
// (low / high half: the arcs that start at k and at k + 8 are computed by the same v_pk_maximum3_f16), and the positions past 7
// of the sliding windows are the SAME registers with their halves exchanged -- op_sel / op_sel_hi of the packed instruction, no
// operation.  8 + 8 v_pk_maximum3_f16 (windows of 3, then 3 + 3 + 3), 5 v_pk_minimum3_f16 over the 16 arcs and
// across the halves.  All values are integers below 2048 in magnitude: exact in f16.  The high halves come from ds_read_u8_d16_hi (on this
// part -- sramecc -- a d16 load ZEROES the other half of its destination, tools/ubench/valu_rate5.hip), the low halves from plain
// byte loads, and the bitop3 that applies bias and sign also merges the two.
#define PG_PK3(OP, d, a, b, c, SEL) asm(OP " %0, %1, %2, %3" SEL : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define PG_SW_NONE ""
#define PG_SW_C " op_sel:[0,0,1] op_sel_hi:[1,1,0]"        // halves of the third source exchanged
#define PG_SW_BC " op_sel:[0,1,1] op_sel_hi:[1,0,0]"       // ... of the second and third
// the exact score of ONE list entry: returns the score when the pixel is a corner at t (score >= t), 0 otherwise; (iy, ix) = its
// interior coordinates
template <bool BFMT>
__device__ __forceinline__ int pk_score_entry(const uint8_t* tile, int e, _Float16 th, int& iy, int& ix)
{
    constexpr int TP = 48;
#define PG_RO(dx, dy) ((3 + (dy)) * TP + 3 + (dx))               // byte offset of ring pixel (dx, dy) from (x - 3, y - 3)
    int bright;
    if (BFMT) bfmt_decode(e, iy, ix, bright);
    else { iy = (e >> 8) & 0x7F; ix = e & 0xFF; bright = e >> 15; }
    const uint8_t* rb = tile + iy * TP + 1 + ix;             // (x - 3, y - 3) of the candidate
    // ring positions 0..7 (OpenCV's order: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)) -> low halves
    const uint32_t l0 = rb[PG_RO(0, 3)], l1 = rb[PG_RO(1, 3)], l2 = rb[PG_RO(2, 2)], l3 = rb[PG_RO(3, 1)];
    const uint32_t l4 = rb[PG_RO(3, 0)], l5 = rb[PG_RO(3, -1)], l6 = rb[PG_RO(2, -2)], l7 = rb[PG_RO(1, -3)];
    const uint32_t v = rb[PG_RO(0, 0)];
    // ... 8..15 ((0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)) -> high halves
    uint32_t h0, h1, h2, h3, h4, h5, h6, h7;
    const uint32_t la = (uint32_t)(uintptr_t)(pg_lptr_t)rb;
    asm volatile("ds_read_u8_d16_hi %0, %8 offset:%9\n\tds_read_u8_d16_hi %1, %8 offset:%10\n\tds_read_u8_d16_hi %2, %8 offset:%11\n\t"
                 "ds_read_u8_d16_hi %3, %8 offset:%12\n\tds_read_u8_d16_hi %4, %8 offset:%13\n\tds_read_u8_d16_hi %5, %8 offset:%14\n\t"
                 "ds_read_u8_d16_hi %6, %8 offset:%15\n\tds_read_u8_d16_hi %7, %8 offset:%16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(h4), "=&v"(h5), "=&v"(h6), "=&v"(h7)
                 : "v"(la), "n"(PG_RO(0, -3)), "n"(PG_RO(-1, -3)), "n"(PG_RO(-2, -2)), "n"(PG_RO(-3, -1)), "n"(PG_RO(-3, 0)),
                   "n"(PG_RO(-3, 1)), "n"(PG_RO(-2, 2)), "n"(PG_RO(-1, 3)));
    const uint32_t K = bright ? 0xE400E400u : 0x64006400u;   // 1024 + byte, negated for the brighter ring
    // (lo | hi) ^ K:  bitop3 0x56
    const uint32_t A0 = __builtin_amdgcn_bitop3_b32(l0, h0, K, 0x56), A1 = __builtin_amdgcn_bitop3_b32(l1, h1, K, 0x56);
    const uint32_t A2 = __builtin_amdgcn_bitop3_b32(l2, h2, K, 0x56), A3 = __builtin_amdgcn_bitop3_b32(l3, h3, K, 0x56);
    const uint32_t A4 = __builtin_amdgcn_bitop3_b32(l4, h4, K, 0x56), A5 = __builtin_amdgcn_bitop3_b32(l5, h5, K, 0x56);
    const uint32_t A6 = __builtin_amdgcn_bitop3_b32(l6, h6, K, 0x56), A7 = __builtin_amdgcn_bitop3_b32(l7, h7, K, 0x56);
    // windows of 3: T_k = (max x[k..k+2], max x[k+8..k+10]); A_8 = A_0 with its halves exchanged, A_9 = A_1 ...
    uint32_t T0, T1, T2, T3, T4, T5, T6, T7;
    PG_PK3("v_pk_maximum3_f16", T0, A0, A1, A2, PG_SW_NONE); PG_PK3("v_pk_maximum3_f16", T1, A1, A2, A3, PG_SW_NONE);
    PG_PK3("v_pk_maximum3_f16", T2, A2, A3, A4, PG_SW_NONE); PG_PK3("v_pk_maximum3_f16", T3, A3, A4, A5, PG_SW_NONE);
    PG_PK3("v_pk_maximum3_f16", T4, A4, A5, A6, PG_SW_NONE); PG_PK3("v_pk_maximum3_f16", T5, A5, A6, A7, PG_SW_NONE);
    PG_PK3("v_pk_maximum3_f16", T6, A6, A7, A0, PG_SW_C);    PG_PK3("v_pk_maximum3_f16", T7, A7, A0, A1, PG_SW_BC);
    // windows of 9: N_k = max(T_k, T_k+3, T_k+6)
    uint32_t N0, N1, N2, N3, N4, N5, N6, N7;
    PG_PK3("v_pk_maximum3_f16", N0, T0, T3, T6, PG_SW_NONE); PG_PK3("v_pk_maximum3_f16", N1, T1, T4, T7, PG_SW_NONE);
    PG_PK3("v_pk_maximum3_f16", N2, T2, T5, T0, PG_SW_C);    PG_PK3("v_pk_maximum3_f16", N3, T3, T6, T1, PG_SW_C);
    PG_PK3("v_pk_maximum3_f16", N4, T4, T7, T2, PG_SW_C);    PG_PK3("v_pk_maximum3_f16", N5, T5, T0, T3, PG_SW_BC);
    PG_PK3("v_pk_maximum3_f16", N6, T6, T1, T4, PG_SW_BC);   PG_PK3("v_pk_maximum3_f16", N7, T7, T2, T5, PG_SW_BC);
    uint32_t r1, r2, r3, r4, M;
    PG_PK3("v_pk_minimum3_f16", r1, N0, N1, N2, PG_SW_NONE); PG_PK3("v_pk_minimum3_f16", r2, N3, N4, N5, PG_SW_NONE);
    PG_PK3("v_pk_minimum3_f16", r3, N6, N7, r1, PG_SW_NONE); PG_PK3("v_pk_minimum3_f16", r4, r2, r3, r3, PG_SW_NONE);
    PG_PK3("v_pk_minimum3_f16", M, r4, r4, r4, PG_SW_C);     // low half: min(low, high)  (the unpacked v_min3_f16 issues at half the rate)
    // sgn * (1024 + v) - 1 as f16 bits: 1023 + v = 0x63FF + v;  -(1025 + v) = 0xE401 + v
    const uint32_t ccb = v + (bright ? 0xE401u : 0x63FFu);
    const _Float16 sh = __builtin_bit_cast(_Float16, (uint16_t)ccb) - __builtin_bit_cast(_Float16, (uint16_t)M);   // the score (OpenCV: arc minimum - 1)
    return sh >= th ? (int)sh : 0;
#undef PG_RO
}

template <bool BFMT>
__device__ __forceinline__ void score_list_pk(const uint8_t* tile, uint8_t* smap, uint16_t* list, int nlist, int t, int lane)
{
    constexpr int MP = 40;
    const _Float16 th = (_Float16)t;
    // (reading the NEXT round's entry a round ahead measured 0.6 % slower: profiles/r05_k2_ab.txt)
    for (int base = 0; base < nlist; base += 64) {
        const int i = base + lane;
        if (i < nlist) {
            int iy, ix;
            const int sc = pk_score_entry<BFMT>(tile, list[i], th, iy, ix);
            const bool corner = sc != 0;
            if (corner) smap[(iy + 1) * MP + ix + 1] = (uint8_t)sc;
            if (BFMT) list[i] = corner ? (uint16_t)((iy << 8) | ix) : (uint16_t)FAST_DEAD;
            else if (!corner) list[i] = (uint16_t)FAST_DEAD;
        }
    }
}

// 3x3 strict NMS of one pixel on the score map (outside the interior = 0); returns its score or 0
__device__ __forceinline__ int nms_score(const uint8_t* smap, int mapPitch, int iy, int ix)
{
    const uint8_t* m = smap + (iy + 1) * mapPitch + ix + 1;
    const int s = m[0];
    // Two-level short circuit: most list entries are not corners (s == 0) and stop after ONE byte read (reading all
    // nine bytes for every lane measured 0.65 ms slower per step in the block form: the phase is bound by LDS round
    // trips per wave, not by LDS throughput); corners read their eight neighbours at once and reduce them with
    // three v_max3 + one v_max instead of eight compare-and-branch steps.
    if (!s) return 0;
    const int a = imax3(m[-mapPitch - 1], m[-mapPitch], m[-mapPitch + 1]);
    const int b = imax3(m[mapPitch - 1], m[mapPitch], m[mapPitch + 1]);
    const int c = imax3(m[-1], m[1], max(a, b));
    return s > c ? s : 0;
}

// ... of a pixel known to be a corner: the nine bytes in one batch
__device__ __forceinline__ int nms_corner(const uint8_t* smap, int mapPitch, int iy, int ix)
{
    const uint8_t* m = smap + (iy + 1) * mapPitch + ix + 1;
    const int s = m[0];
    const int a = imax3(m[-mapPitch - 1], m[-mapPitch], m[-mapPitch + 1]);
    const int b = imax3(m[mapPitch - 1], m[mapPitch], m[mapPitch + 1]);
    const int c = imax3(m[-1], m[1], max(a, b));
    return s > c ? s : 0;
}

// Slow path for one threshold when the candidate list would not fit in LDS (very noisy cell at
// minThFAST): score the cell in row blocks, then run NMS over every interior pixel.
// Leaves the score map zeroed.  Returns the number of survivors (written to `out`).
__device__ __noinline__ int fast_pass_chunked(const uint8_t* tile, int TP, uint8_t* smap, int mapPitch,
                                              int mapRows, int IW, int IH, int t, uint16_t* list,
                                              uint32_t* out, int cap, int xoff, int yoff, int lane)
{
    PG_WAVE_SYNC();
    for (int i = lane; i < (mapRows * mapPitch) >> 2; i += 64) reinterpret_cast<uint32_t*>(smap)[i] = 0;
    PG_WAVE_SYNC();
    const int rowsPer = (IW <= 32) ? 16 : 8;                 // <= 512 candidates per block
    for (int r = 0; r < IH; r += rowsPer) {
        const int n = (IW <= 32) ? quick_pass<8, true>(tile, TP, IW, r, min(r + rowsPer, IH), t, list, lane)
                                 : quick_pass<16, true>(tile, TP, IW, r, min(r + rowsPer, IH), t, list, lane);
        PG_WAVE_SYNC();
        score_list<false>(tile, TP, smap, mapPitch, list, n, t, lane);
        PG_WAVE_SYNC();
    }
    int done = 0;
    for (int iy = 0; iy < IH; iy++)
        for (int ix = lane; ix - lane < IW; ix += 64) {
            const int sc = ix < IW ? nms_score(smap, mapPitch, iy, ix) : 0;
            const unsigned long long m = __ballot(sc != 0);
            if (sc) {
                const int pos = done + wave_prefix(m);
                if (pos < cap) out[pos] = (uint32_t)(ix + xoff) | ((uint32_t)(iy + yoff) << 12) | ((uint32_t)sc << 24);
            }
            done += __popcll(m);
        }
    PG_WAVE_SYNC();
    for (int i = lane; i < (mapRows * mapPitch) >> 2; i += 64) reinterpret_cast<uint32_t*>(smap)[i] = 0;
    PG_WAVE_SYNC();
    return done;
}


#ifdef PGORB_FAST_TIMING
// developer build only (make EXTRA=-DPGORB_FAST_TIMING): 10 ns ticks per phase of every wave (no
// atomics: a shared counter would serialise the waves), read back by tools/experiments/fast_timing.py
#define FT_MAXW (1 << 20)
__device__ unsigned int pg_ft_log[FT_MAXW * 8];
#define FT_TS(k) do { const unsigned long long t1_ = wall_clock64(); if (lane == 0 && ft_id < FT_MAXW) pg_ft_log[ft_id * 8 + (k)] = (unsigned)(t1_ - ft_t0); ft_t0 = t1_; } while (0)
extern "C" int pgorb_debug_fast_times(unsigned int* out, int nwaves)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pg_ft_log), sizeof(unsigned) * 8 * (size_t)nwaves) == hipSuccess ? 0 : -1;
}
#define FT_PARAMS , unsigned long long& ft_t0, const int ft_id
#define FT_ARGS , ft_t0, ft_id
#else
#define FT_TS(k) do {} while (0)
#define FT_PARAMS
#define FT_ARGS
#endif

// One detector pass at threshold t over the staged window == cv::FAST(window, t, true): necessary test + compaction,
// exact scores, NMS, survivors into the cell's slots.  Returns the number of survivors.  STRONG: the four-pair test
// (the minThFAST retry).
template <int TPC, int MPC, bool NARROW, bool STRONG>
__device__ __forceinline__ int fast_pass(int32_t* status, const uint8_t* tile, int TP, uint8_t* smap, int mapPitch, int mapRows,
                                         int IW, int IH, int t, uint16_t* list, uint32_t* out, int cellCap, int xoff, int yoff, int lane,
                                         const PgLaneValid& V FT_PARAMS)
{
    // (2) necessary test + compaction
    const int nlist = NARROW ? quick_pass_b<STRONG>(tile, IW, IH, t, list, lane, V)
                    : (IW <= 32) ? quick_pass<8, STRONG>(tile, TP, IW, 0, IH, t, list, lane)
                                 : quick_pass<16, STRONG>(tile, TP, IW, 0, IH, t, list, lane);
#if defined(PGORB_FAST_STOP) && PGORB_FAST_STOP == 3        // developer builds (tools/experiments/r5_k2_stages.sh): stop behind a stage,
    return nlist > 0 ? 0x40000000 | nlist : 0;              // with the stage's result kept alive
#endif
#if defined(PGORB_FAST_STOP) && PGORB_FAST_STOP == 2
    return nlist;
#endif
    if (nlist < 0)                                         // list would overflow: chunked slow path
        return fast_pass_chunked(tile, TP, smap, mapPitch, mapRows, IW, IH, t, list, out, cellCap, xoff, yoff, lane);
    PG_WAVE_SYNC();
    if (!STRONG) FT_TS(5);
    if (TPC == 48 && MPC == 40 && !PG_FAST_SCORE_I32 && !PG_FAST_NO_FUSED && nlist <= 64) {
        // ONE round (most cells at iniThFAST): the lane that scored an entry also suppresses and emits it -- the score stays in its
        // register, so NMS starts at the neighbour reads: no re-filed list entry, no list read, no centre read (two dependent LDS
        // round trips and a dozen instructions of a wave whose life is its instruction count)
        int sc = 0, iy = 0, ix = 0;
        if (lane < nlist) {
            sc = pk_score_entry<NARROW>(tile, list[lane], (_Float16)t, iy, ix);
            if (sc) smap[(iy + 1) * 40 + ix + 1] = (uint8_t)sc;
        }
        PG_WAVE_SYNC();
        if (!STRONG) FT_TS(6);
        if (sc) {
            const uint8_t* m = smap + (iy + 1) * 40 + ix + 1;
            const int a = imax3(m[-41], m[-40], m[-39]);
            const int b = imax3(m[39], m[40], m[41]);
            const int c = imax3(m[-1], m[1], max(a, b));
            sc = sc > c ? sc : 0;
        }
        const unsigned long long mk = __ballot(sc != 0);
        if (sc) {
            const uint32_t pos = (uint32_t)wave_prefix(mk);
            if (pos < (uint32_t)cellCap)
                out[pos] = (uint32_t)(ix + xoff) | ((uint32_t)(iy + yoff) << 12) | ((uint32_t)sc << 24);
            else
                atomicExch(status, PGORB_E_OVERFLOW);            // cannot happen (see header)
        }
        return __popcll(mk);
    }
    // (3) exact scores for the compacted pixels
    if (TPC == 48 && MPC == 40 && !PG_FAST_SCORE_I32) score_list_pk<NARROW>(tile, smap, list, nlist, t, lane);
    else score_list<NARROW>(tile, TP, smap, mapPitch, list, nlist, t, lane);
    PG_WAVE_SYNC();
    if (!STRONG) FT_TS(6);
#if defined(PGORB_FAST_STOP) && PGORB_FAST_STOP == 4
    return 0x40000000 | (nlist ? list[lane % nlist] : 0);
#endif
    // (4) NMS (strictly greater than all 8 neighbours; outside the interior = 0); survivors go
    // straight into this cell's slots
    int total = 0;
    for (int base = 0; base < nlist; base += 64) {
        const int i = base + lane;
        int sc = 0, p = 0;
        if (i < nlist) {
            p = list[i];
            // (dead: not a corner at t.)  Every entry that is still alive IS a corner -- the scoring round marked the others -- so its
            // own score and its eight neighbours are read in ONE batch: the short circuit on the centre byte that nms_score
            // keeps for the row-chunked path would be a dependent LDS round trip per round here
            if (p != (int)FAST_DEAD) sc = nms_corner(smap, mapPitch, (p >> 8) & 0x7F, p & 0xFF);
        }
        const unsigned long long m = __ballot(sc != 0);
        if (sc) {
            const uint32_t pos = (uint32_t)(total + wave_prefix(m));      // (unsigned: scalar base + 32-bit lane offset, no 64-bit address arithmetic)
            const int iy = (p >> 8) & 0x7F, ix = p & 0xFF;
            if (pos < (uint32_t)cellCap)
                out[pos] = (uint32_t)(ix + xoff) | ((uint32_t)(iy + yoff) << 12) | ((uint32_t)sc << 24);
            else
                atomicExch(status, PGORB_E_OVERFLOW);            // cannot happen (see header)
        }
        total += __popcll(m);
    }
    return total;
}

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
#ifdef PGORB_FAST_TIMING
    unsigned long long ft_t0 = wall_clock64();
    const int ft_id = (int)frame * totalCells + cell;
#endif
    const uint32_t* recp = tab + 16 * (int64_t)cell;               // (the tables have 8 records of slack)
    pg_u32x16 rec;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rec) : "s"(recp) : "memory");
    if (cell >= cellEnd || slot >= cellsPerXcd) return cpw;
    const int iniX = rec[1] & 0xFFFF, iniY = rec[1] >> 16;
    const int W = rec[2] & 0xFF, H = (rec[2] >> 8) & 0xFF, cellCap = rec[2] >> 17;
    int32_t* cellCnt = cellCountBase + ((uint64_t)frame * (uint32_t)totalCells + (rec[0] >> 4));     // the record names its cell
    FT_TS(0);
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
#ifdef PGORB_FAST_TIMING
        asm volatile("" :: "v"(voff));
        FT_TS(4);
#endif
        const int nz = (mapRows * mapPitch + 15) >> 4;
        const bool laneOn = r0 < rowsPer;
        // Round 5: in the common geometry the first TWO loads (rows 0-20, 21-41) are issued by every lane with a chunk, whatever the
        // window's height: a row past the window is CLAMPED to its last row (a duplicate read of a line that is being fetched anyway), so
        // there is no guard per instruction -- three of them cost 36 scalar instructions and three exec round trips on a wave whose life
        // is its instruction count.  Rows 37-41 of a 37-row window land in the first bytes of the score map, which is therefore cleared
        // AFTER the landing now (below); a third load only exists for windows taller than 42 rows (wave-uniform branch).
        constexpr bool CLAMPED = TPC == 48 && MPC == 40 && NARROW && !PG_FAST_NO_CLAMP;
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
        FT_TS(1);
        __builtin_amdgcn_s_waitcnt(0);                     // vmcnt(0): the DMA has landed
        if (CLAMPED) {                                     // <= 42 rows of 40 bytes: two unconditional steps of 1 KiB -- what they clear
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);    // past the map is the candidate list, written later (the launcher keeps
            reinterpret_cast<uint4*>(smap)[lane] = z;      // map + list >= 2 KiB)
            reinterpret_cast<uint4*>(smap)[lane + 64] = z;
        }
    }
    PG_WAVE_SYNC();
    FT_TS(2);
    // developer builds (tools/experiments/r5_k2_sensitivity.sh): what one more unit of each resource costs the launch -- n extra
    // fast-class VALU instructions, n scalar instructions, n conflict-free LDS reads or n x 64 cycles of sleep per cell
#ifdef PGORB_FAST_PAD_VALU
    { uint32_t z = lane;
#pragma unroll
      for (int i = 0; i < PGORB_FAST_PAD_VALU; i++) asm volatile("v_add_u32_e32 %0, %0, %0" : "+v"(z));
      asm volatile("" :: "v"(z)); }
#endif
#ifdef PGORB_FAST_PAD_SALU
    { uint32_t z = (uint32_t)cell;
#pragma unroll
      for (int i = 0; i < PGORB_FAST_PAD_SALU; i++) asm volatile("s_add_u32 %0, %0, %0" : "+s"(z) :: "scc");
      asm volatile("" :: "s"(z)); }
#endif
#ifdef PGORB_FAST_PAD_LDS
    { uint32_t z = 0; const uint32_t* tw = reinterpret_cast<const uint32_t*>(tile) + lane;
#pragma unroll
      for (int i = 0; i < PGORB_FAST_PAD_LDS; i++) z += tw[(i & 7) * 64];
      asm volatile("" :: "v"(z)); }
#endif
#ifdef PGORB_FAST_PAD_SLEEP
#pragma unroll
    for (int i = 0; i < PGORB_FAST_PAD_SLEEP; i++) __builtin_amdgcn_s_sleep(1);
#endif
#if defined(PGORB_FAST_STOP) && PGORB_FAST_STOP == 1       // window staged, nothing else
    if (lane == 0) *cellCnt = reinterpret_cast<const uint32_t*>(tile)[17] & 1;
    return cpw;
#endif

    uint32_t* out = cellCandBase + ((uint64_t)frame * cellCandFrame + rec[7]);
    const PgCellValid cv = {rec[8], rec[9], rec[10], rec[11], rec[12], rec[13], rec[14], rec[15]};
    const PgLaneValid valid = NARROW ? pg_lane_valid(cv) : PgLaneValid{0u, 0u};
    const int xoff = 3 + iniX - PG_EDGE, yoff = 3 + iniY - PG_EDGE;   // window-local -> region-relative (:822-823)

    // The two detector passes written out (round 3): as a `for (pass)` loop the compiler merged the two bodies and paid for it
    // with scalar flag juggling around every phase.
    int total = fast_pass<TPC, MPC, NARROW, false>(statusPtr, tile, TP, smap, mapPitch, mapRows, IW, IH, iniTh, list, out, cellCap, xoff, yoff, lane, valid FT_ARGS);
    FT_TS(7);
#if defined(PGORB_FAST_SKIP) || defined(PGORB_FAST_STOP)   // timing experiments: no minTh retry
    if (lane == 0) *cellCnt = min(total & 0xFFFF, cellCap);
    return cpw;
#endif
    if (total == 0) {
        // vKeysCell.empty() -> retry at minThFAST (:812-816).  The score map keeps what the first pass wrote: a FAST score does
        // not depend on the threshold and every corner at iniThFAST is a candidate of the retry again (an "empty" cell
        // can hold corners -- equal neighbouring maxima that strict NMS removed)
        PG_WAVE_SYNC();
        total = fast_pass<TPC, MPC, NARROW, true>(statusPtr, tile, TP, smap, mapPitch, mapRows, IW, IH, minTh, list, out, cellCap, xoff, yoff, lane, valid FT_ARGS);
    }
    if (lane == 0) *cellCnt = min(total, cellCap);
    FT_TS(3);
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

// describe.hip -- K4+K5+K6 fused: orientation, 7x7 Gaussian blur and rBRIEF-256, one
// 64-lane wave per keypoint (gfx950).
//
// Restates, for every keypoint the quadtree kept:
//   IC_Angle / computeOrientation   thirdparty/orb-slam2/src/ORBextractor.cc:77-104, 472-479
//   GaussianBlur(7x7, sigma 2, REFLECT_101) of the level, ORBextractor.cc:1084-1085
//   computeOrbDescriptor            ORBextractor.cc:107-147 with bit_pattern_31_ :150-408
//   output assembly (pt *= scale, octave, size)   ORBextractor.cc:836-846, 1094-1102
//
// The reference blurs every full pyramid level; only a 37x37 neighbourhood of each keypoint
// is ever sampled (rotated taps reach +-18 px), so the wave stages the 43x43 RAW window in
// LDS once (LDS-DMA; reflect-101 addressing only for windows that touch the image edge),
// takes the integer moments from it, runs the ROW pass of the separable fixed-point blur
// LDS->LDS and the COLUMN pass only at the 512 rotated tap positions.  The 256 comparisons are packed
// with four 64-bit wave ballots: ballot bit j of round r is descriptor bit 64r+j, which is
// exactly the reference's LSB-first byte packing (:127-141).
//
// Float steps that decide an integer (tap coordinates) use explicit round-to-nearest
// intrinsics, never contracted to FMA, and the sin/cos pair is evaluated in double by a fixed
// sequence and rounded once (parity contract, see oracle/orb_oracle.c orc_sincos_f).
// Algorithmic bytes per keypoint: 43*43 window read + 60 B written.
#include "pgorb_internal.h"

// bit_pattern_31_ (ORBextractor.cc:150-408) as floats: the test points are only ever used as
// float factors of the rotation (:119-120), so the table holds them converted once at compile time
struct PgPatternF { float v[256 * 4]; };
constexpr PgPatternF pg_make_pattern_f()
{
    constexpr int8_t p[256 * 4] = {
#include "orb_pattern31.inc"
    };
    PgPatternF t = {};
    for (int i = 0; i < 256 * 4; i++) t.v[i] = (float)p[i];
    return t;
}
__device__ static const PgPatternF pg_pattern31f = pg_make_pattern_f();

// (circular patch row half-widths umax, ORBextractor.cc:452-469, are baked into pg_make_moment_tab)

struct PgMomentTab { uint32_t wu[31 * 9], wm[31 * 9]; };
// per (disc row v+15, window dword q+1): packed byte weights (u+15 inside the disc, else 0) and
// the 0/1 disc mask, for v_dot4_u32_u8
constexpr PgMomentTab pg_make_moment_tab()
{
    constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    PgMomentTab t = {};
    for (int r = 0; r < 31; r++)
        for (int q = 0; q < 9; q++) {
            const int vy = r - 15, dmax = umax[vy < 0 ? -vy : vy];
            uint32_t wu = 0, wm = 0;
            for (int k = 0; k < 4; k++) {
                const int ux = 4 * (q + 1) + k - 21;
                if ((ux < 0 ? -ux : ux) <= dmax) { wu |= (uint32_t)(ux + 15) << (8 * k); wm |= 1u << (8 * k); }
            }
            t.wu[r * 9 + q] = wu; t.wm[r * 9 + q] = wm;
        }
    return t;
}
__device__ static const PgMomentTab pg_moment_tab = pg_make_moment_tab();

#define DW_R 21               // window radius: 18 (taps) + 3 (blur)
#define DW_N 43               // window side
#define DW_PITCH 48
#define DH_N 37               // blurred side
#define DH_PITCH 40           // columns per packed row pair of the horizontal pass
#define DB_PITCH 40

__device__ __forceinline__ int pg_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}

// cv::fastAtan2 (OpenCV 2.4 core/mathfuncs.cpp), unfused single precision.
__device__ __forceinline__ float pg_fast_atan2(float y, float x)
{
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sin/cos in IEEE double by a fixed operation sequence, rounded once to float.
__device__ __forceinline__ void pg_sincos_f(float angle, float* s_out, float* c_out)
{
    const double INV_PIO2 = 0.63661977236758138243;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double x = (double)angle;
    const double kd = floor(__dadd_rn(__dmul_rn(x, INV_PIO2), 0.5));
    const double r = __dsub_rn(__dsub_rn(x, __dmul_rn(kd, PIO2_HI)), __dmul_rn(kd, PIO2_LO));
    const double r2 = __dmul_rn(r, r);
    double ps = -1.0 / 355687428096000.0;
    ps = __dadd_rn(__dmul_rn(ps, r2), 1.0 / 1307674368000.0);
    ps = __dsub_rn(__dmul_rn(ps, r2), 1.0 / 6227020800.0);
    ps = __dadd_rn(__dmul_rn(ps, r2), 1.0 / 39916800.0);
    ps = __dsub_rn(__dmul_rn(ps, r2), 1.0 / 362880.0);
    ps = __dadd_rn(__dmul_rn(ps, r2), 1.0 / 5040.0);
    ps = __dsub_rn(__dmul_rn(ps, r2), 1.0 / 120.0);
    ps = __dadd_rn(__dmul_rn(ps, r2), 1.0 / 6.0);
    const double sn = __dsub_rn(r, __dmul_rn(__dmul_rn(r, r2), ps));
    double pc = -1.0 / 6402373705728000.0;
    pc = __dadd_rn(__dmul_rn(pc, r2), 1.0 / 20922789888000.0);
    pc = __dsub_rn(__dmul_rn(pc, r2), 1.0 / 87178291200.0);
    pc = __dadd_rn(__dmul_rn(pc, r2), 1.0 / 479001600.0);
    pc = __dsub_rn(__dmul_rn(pc, r2), 1.0 / 3628800.0);
    pc = __dadd_rn(__dmul_rn(pc, r2), 1.0 / 40320.0);
    pc = __dsub_rn(__dmul_rn(pc, r2), 1.0 / 720.0);
    pc = __dadd_rn(__dmul_rn(pc, r2), 1.0 / 24.0);
    pc = __dsub_rn(__dmul_rn(pc, r2), 0.5);
    const double cs = __dadd_rn(1.0, __dmul_rn(r2, pc));
    const int k = (int)((long long)kd & 3);
    const double s = (k == 0) ? sn : (k == 1) ? cs : (k == 2) ? -sn : -cs;
    const double c = (k == 0) ? cs : (k == 1) ? -sn : (k == 2) ? -cs : sn;
    *s_out = (float)s;
    *c_out = (float)c;
}

struct PgGauss7 { int k0, k1, k2, k3; };     // K[0]=K[6]=k0 ... K[3]=k3 (8-bit fixed point)

// sum over the 64 lanes, returned wave-uniform: DPP row shifts inside each row of 16, row
// broadcasts across rows, then lane 63 (six cross-lane moves instead of six LDS-crossbar shuffles)
__device__ __forceinline__ int pg_wave_sum(int x)
{
    int v = x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}


typedef __attribute__((address_space(1))) const void* pg_gptr_t;
typedef __attribute__((address_space(3))) void* pg_lptr_t;
typedef unsigned short pg_us2 __attribute__((ext_vector_type(2)));
typedef int pg_v4i __attribute__((ext_vector_type(4)));

// B operand of the row-pass MFMAs: the banded Toeplitz matrix of the 7 taps as i8, [column block
// cb][lane l][16 bytes]: byte j of an entry is T[k][n] = K[k - n] (0 <= k - n <= 6, n < 37) for
// k = 16 (l >> 4) + j, n = 16 cb + (l & 15).  Filled once by pg_launch_describe.
__device__ uint32_t pg_blur_btab[3 * 64 * 4];

// Column pass of the separable blur for ONE output pixel (Y, X) of the 37x37 blurred tile, from
// the row-pass sums hT[row pair][column] = (row 2p | row 2p+1 << 16): rows Y .. Y+6 are three
// packed pairs and a single, shifted by 16 bits when Y is odd.  FixedPtCastEx rounding (half
// up) and the SSE2 tie rule of the reference build (tie -> even for x < (w & ~3)).
__device__ __forceinline__ int pg_blur_at(const uint32_t* hT, int Y, int X, const PgGauss7& G,
                                          pg_us2 K01, pg_us2 K23, pg_us2 K45, bool tieEven)
{
    const uint32_t* s = hT + (Y >> 1) * 40 + X;
    const uint32_t p0 = s[0], p1 = s[40], p2 = s[80], p3 = s[120];
    const uint32_t sh = (uint32_t)(Y & 1) << 4;
    const uint32_t a0 = __builtin_amdgcn_alignbit(p1, p0, sh), a1 = __builtin_amdgcn_alignbit(p2, p1, sh),
                   a2 = __builtin_amdgcn_alignbit(p3, p2, sh), last = (p3 >> sh) & 0xFFFFu;
    uint32_t C = __builtin_amdgcn_udot2(__builtin_bit_cast(pg_us2, a0), K01, (uint32_t)G.k0 * last, false);
    C = __builtin_amdgcn_udot2(__builtin_bit_cast(pg_us2, a1), K23, C, false);
    C = __builtin_amdgcn_udot2(__builtin_bit_cast(pg_us2, a2), K45, C, false);
    int v = (int)((C + 32768u) >> 16);
    if (tieEven && (C & 0xFFFFu) == 0x8000u) v &= ~1;
    return min(v, 255);
}

// wave-local "barrier": a k_describe workgroup is ONE wave, LDS operations of a wave execute in order, so the only
// thing to enforce is that the compiler keeps LDS accesses on their side of the point.  __syncthreads() costs a full
// s_waitcnt vmcnt(0) lgkmcnt(0) each time (K2: 4 % of the kernel).
#define PG_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__global__ __launch_bounds__(64) void k_describe(const PgPlan P, const PgGauss7 G,
                                                  pgorb_keypoint* __restrict__ kps,
                                                  uint8_t* __restrict__ desc, int cap_per_frame,
                                                  int32_t* __restrict__ n_out, int slotBeg, int slotEnd, int writeTotal, int argPad)
{
    // one 3520-byte LDS buffer per wave: first the raw window (43 rows x 48 B), then -- once the
    // moments and the row-pass operands have been read from it -- the row sums [row pair][column]
    // (8 waves per SIMD instead of 7: the wave is a long dependent chain, occupancy is throughput)
    __shared__ __attribute__((aligned(16))) uint32_t hbuf[22 * DH_PITCH];
    uint8_t* raw = reinterpret_cast<uint8_t*>(hbuf);

    const int lane = threadIdx.x;
    const int frame = blockIdx.y;
    // XCD-contiguous keypoint ranges: consecutive workgroups go to consecutive XCDs, so give XCD x
    // the x-th eighth of the frame's keypoint list (neighbours in the list are neighbours in the
    // image: their 43x43 windows share L2 lines)
    // (the launch covers slots [slotBeg, slotEnd) -- the slabs of a range of levels; writeTotal: this launch reports the
    //  frame's keypoint count, i.e. K3 has finished for every level)
    const int slot = slotBeg + (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // The wave is one long dependent chain and lives ~8 us; every load that does not depend on
    // the keypoint is issued here, before the chain starts: the lane's entries of the moment
    // table, the Toeplitz operands of the row pass and its four test-point pairs.
    uint32_t mtU[5], mtM[5];
#pragma unroll
    for (int it = 0; it < 5; it++) {
        const int i = min(lane + 64 * it, 31 * 9 - 1);
        mtU[it] = pg_moment_tab.wu[i]; mtM[it] = pg_moment_tab.wm[i];
    }
    pg_v4i Bop[3];
#pragma unroll
    for (int cb = 0; cb < 3; cb++) Bop[cb] = reinterpret_cast<const pg_v4i*>(pg_blur_btab)[cb * 64 + lane];
    float4 pat[4];
#pragma unroll
    for (int r = 0; r < 4; r++) pat[r] = *reinterpret_cast<const float4*>(pg_pattern31f.v + 4 * (64 * r + lane));
    // Slot s of a frame IS entry s of the frame's selection slab (the levels' slabs lie back to back, K3 writes them in
    // dispatch order), and the 32-byte record holds the keypoint AND what the wave needs of its level (PgSelRec): the
    // record and the frame's per-level counts load in ONE scalar trip behind the arguments.  (The first version searched
    // the counts for the slot's level, loaded the record, then the level's fields: three dependent round trips were
    // 2.6 of the wave's 8.1 us.)  A level that kept fewer keypoints than its capacity (~1 % of the slots) leaves marked
    // records behind; their waves return after the loads.
    const int32_t* kpc = P.kpCount + frame * PG_MAXL;
    const int nlevels = P.nlevels;
    const PgSelRec* recp = reinterpret_cast<const PgSelRec*>(P.sel) + ((int64_t)frame * P.selFrame + slot);
    const int tieMode = P.tieMode;
    asm volatile("" :: "s"(kpc), "s"(nlevels), "s"(recp), "s"(n_out), "s"(kps), "s"(desc), "s"(cap_per_frame), "s"(tieMode),
                 "s"(slotBeg), "s"(slotEnd), "s"(writeTotal), "s"(argPad));     // (argPad completes the 16-byte group of the three ints: with
                 // a dead fourth dword the allocator reused its register for another argument load and split the batch in two)
    typedef int32_t pg_i32x16 __attribute__((ext_vector_type(16)));
    typedef uint32_t pg_u32x8 __attribute__((ext_vector_type(8)));
    pg_i32x16 kc;
    pg_u32x8 rec;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(kc), "=&s"(rec) : "s"(kpc), "s"(recp) : "memory");
    if (slot >= slotEnd) return;                            // (grid padding; never the launch's first slot)
    const int l = (int)(rec[1] >> 16) & 15;                 // (an unused slot's level is masked here and refused below)
    int total = 0, before = 0;                              // keypoints of the frame / of the levels below l
#pragma unroll
    for (int q = 0; q < PG_MAXL; q++) {
        const int c = kc[q];                                // (levels the plan does not have: zero, run_batch clears the counters and K3 never writes them)
        if (q < l) before += c;
        total += c;
    }
    if (writeTotal && blockIdx.x == 0 && lane == 0) n_out[frame] = total;      // (the launch's first slot)
    if (rec[1] == 0xFFFFFFFFu || l >= nlevels) return;      // an unused slot of a level's slab (K3 marks them)
    const uint32_t cv = rec[0];
    const int idx = before + (int)(rec[1] & 0xFFFFu);       // the OUTPUT slot is the keypoint's position in the reference's list
    const int Lpitch = (int)rec[2], Lw = (int)(rec[3] & 0xFFFFu), Lh = (int)(rec[3] >> 16);
    const uint8_t* img = reinterpret_cast<const uint8_t*>((uintptr_t)(((uint64_t)rec[5] << 32) | rec[4]));
    const float Lscale = __uint_as_float(rec[6]), LpatchSize = __uint_as_float(rec[7]);
    const int x = (int)(cv & 0xFFF) + PG_EDGE, y = (int)((cv >> 12) & 0xFFF) + PG_EDGE;   // :842-843
    const int resp = (int)(cv >> 24);
    const int w = Lw, h = Lh;

    // ---- stage the raw 43x43 window: window column 0 lands on an LDS dword boundary --------
    const int x0 = x - DW_R, y0 = y - DW_R;
    if (x0 >= 0 && y0 >= 0 && x0 + DW_PITCH <= w && y + DW_R < h) {   // 48 bytes stay inside the row
        // LDS-DMA, 16 B per lane, byte-unaligned global addresses (tools/ubench/glds_unaligned.hip):
        // 3 lanes per 48-byte row, 21 rows per instruction
        const int lr = lane / 3, lq = lane - 3 * lr;
        const uint8_t* g = img + (int64_t)(y0 + lr) * Lpitch + x0 + 16 * lq;
        if (lr < 21) {
            __builtin_amdgcn_global_load_lds((pg_gptr_t)g, (pg_lptr_t)raw, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((pg_gptr_t)(g + 21 * (int64_t)Lpitch), (pg_lptr_t)(raw + 21 * DW_PITCH), 16, 0, 0);
            if (lr < 1)
                __builtin_amdgcn_global_load_lds((pg_gptr_t)(g + 42 * (int64_t)Lpitch), (pg_lptr_t)(raw + 42 * DW_PITCH), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
    } else {                                   // BORDER_REFLECT_101 (:1085)
        for (int i = lane; i < DW_N * DW_PITCH; i += 64) {
            const int r = i / DW_PITCH, c = i - r * DW_PITCH;
            raw[r * DW_PITCH + c] =
                img[(int64_t)pg_reflect101(y0 + r, h) * Lpitch + pg_reflect101(x0 + c, w)];
        }
    }
    PG_WAVE_SYNC();

    // ---- IC_Angle: integer moments over the radius-15 disc (:77-104) -------------------
    // One task = one aligned dword (4 pixels) of one disc row; v_dot4_u32_u8 against the
    // precomputed per-(row, dword) weights: (u+15) inside the disc / 0 outside, and the disc mask.
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int it = 0; it < 5; it++) {
        const int i = lane + 64 * it;
        if (i >= 31 * 9) break;
        const int r = i / 9, q = i - r * 9;                     // r = v+15, dword q+1 = columns 4q+4 ..
        const uint32_t val = *reinterpret_cast<const uint32_t*>(raw + (DW_R - 15 + r) * DW_PITCH + 4 * (q + 1));
        const int sv = (int)__builtin_amdgcn_udot4(val, mtM[it], 0u, false);                   // sum of disc pixels
        m10 += (int)__builtin_amdgcn_udot4(val, mtU[it], 0u, false) - 15 * sv;                 // sum u * I
        m01 += (r - 15) * sv;                                                                   // sum v * I
    }
    m10 = pg_wave_sum(m10);
    m01 = pg_wave_sum(m01);
    const float angle = pg_fast_atan2((float)m01, (float)m10);

    // ---- 7x7 Gaussian, fixed point, separable (OpenCV 2.4 8U path, Appendix A4) ---------
    // row pass on the matrix cores: rowsum[r][n] = sum_t K[t] * raw[r][n + t] is the product of the
    // window (48 x 64, u8 taken as i8 by flipping bit 7, i.e. minus 128) with the banded Toeplitz
    // matrix of the taps (64 x 48, pg_blur_btab): nine v_mfma_i32_16x16x64_i8, exact in int32,
    // with 128 * sum(K) as the accumulator's start value.  Columns 43.. of a row operand are the
    // next row's bytes (whatever they are: the matching Toeplitz rows are zero), rows 43.. are
    // never stored.  The sums of two rows are stored packed (row r | row r+1 << 16) so the column
    // pass can use v_dot2_u32_u16.  (The VALU version was 185 instructions per keypoint.)
    uint32_t* hT = hbuf;                                       // [22 row pairs][DH_PITCH columns]
    {
        const int cinit = 128 * (2 * (G.k0 + G.k1 + G.k2) + G.k3);
        pg_v4i Aop[3];
        // ALL row operands leave the raw window before the first row sum is written over it
#pragma unroll
        for (int rb = 0; rb < 3; rb++) {
            Aop[rb] = *reinterpret_cast<const pg_v4i*>(raw + (16 * rb + (lane & 15)) * DW_PITCH + 16 * (lane >> 4));
            Aop[rb].x ^= (int)0x80808080; Aop[rb].y ^= (int)0x80808080; Aop[rb].z ^= (int)0x80808080; Aop[rb].w ^= (int)0x80808080;
        }
        PG_WAVE_SYNC();
#pragma unroll
        for (int rb = 0; rb < 3; rb++) {
            const int pair = 8 * rb + 2 * (lane >> 4);
#pragma unroll
            for (int cb = 0; cb < 3; cb++) {
                const pg_v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aop[rb], Bop[cb], pg_v4i{cinit, cinit, cinit, cinit}, 0, 0, 0);
                const int n = 16 * cb + (lane & 15);
                if (n < DH_PITCH) {                             // each sum <= 257*255 = 65535
                    if (pair < 22) hT[pair * DH_PITCH + n] = (uint32_t)acc.x | ((uint32_t)acc.y << 16);
                    if (pair + 1 < 22) hT[(pair + 1) * DH_PITCH + n] = (uint32_t)acc.z | ((uint32_t)acc.w << 16);
                }
            }
        }
    }
    PG_WAVE_SYNC();
    // column pass: on demand.  Only the 512 rotated tap positions of the 37x37 blurred tile are
    // ever read, so each lane blurs its own 8 taps from the row-pass sums (4 LDS dwords, 3
    // v_dot2_u32_u16 each) instead of the wave producing all 1369 pixels.
    const pg_us2 K01 = __builtin_bit_cast(pg_us2, (uint32_t)G.k0 | ((uint32_t)G.k1 << 16));
    const pg_us2 K23 = __builtin_bit_cast(pg_us2, (uint32_t)G.k2 | ((uint32_t)G.k3 << 16));
    const pg_us2 K45 = __builtin_bit_cast(pg_us2, (uint32_t)G.k2 | ((uint32_t)G.k1 << 16));
    const int wvec = w & ~3;

    // ---- rBRIEF-256 (:107-147) ------------------------------------------------------------
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    pg_sincos_f(__fmul_rn(angle, factorPI), &b, &a);
    unsigned long long bits[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float px0 = pat[r].x, py0 = pat[r].y, px1 = pat[r].z, py1 = pat[r].w;
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(px0, b), __fmul_rn(py0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(px0, a), __fmul_rn(py0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(px1, b), __fmul_rn(py1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(px1, a), __fmul_rn(py1, b)));
        const int t0 = pg_blur_at(hT, 18 + r0, 18 + c0, G, K01, K23, K45, tieMode == 0 && x + c0 < wvec);
        const int t1 = pg_blur_at(hT, 18 + r1, 18 + c1, G, K01, K23, K45, tieMode == 0 && x + c1 < wvec);
        bits[r] = __ballot(t0 < t1);
    }

    // ---- outputs (:836-846, :1094-1102) -----------------------------------------------------
    if (idx < cap_per_frame && lane == 0) {
        const int64_t o = (int64_t)frame * cap_per_frame + idx;
        unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + o * 32);
        d[0] = bits[0]; d[1] = bits[1]; d[2] = bits[2]; d[3] = bits[3];
        pgorb_keypoint k;
        k.x = (l != 0) ? __fmul_rn((float)x, Lscale) : (float)x;
        k.y = (l != 0) ? __fmul_rn((float)y, Lscale) : (float)y;
        k.size = LpatchSize;
        k.angle = angle;
        k.response = (float)resp;
        k.octave = l;
        k.class_id = -1;
        kps[o] = k;
    }
}

static PgGauss7 pg_gauss7()
{
    // cv::getGaussianKernel(7, 2, CV_32F) then Mat::convertTo(CV_32S, 256) (Appendix A4)
    float cf[7];
    double sum = 0;
    for (int i = 0; i < 7; i++) {
        const double xx = i - 3.0;
        cf[i] = (float)exp(-0.5 / (2.0 * 2.0) * xx * xx);
        sum += cf[i];
    }
    sum = 1. / sum;
    int K[7];
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        K[i] = (int)lrint((double)(cf[i] * 256.f));
    }
    PgGauss7 g = {K[0], K[1], K[2], K[3]};
    return g;
}

void pg_launch_describe(const PgPlan& P, int nframes, pgorb_keypoint* d_kps, uint8_t* d_desc,
                        int cap_per_frame, int32_t* d_n, hipStream_t s)
{
    pg_launch_describe_levels(P, nframes, d_kps, d_desc, cap_per_frame, d_n, 0, P.nlevels, s);
}

// K4-6 for the keypoints of levels [levelBeg, levelEnd): their slots are one contiguous range of a frame's selection slab,
// and a keypoint's output position needs the counts of the levels BELOW its own only.  The launch that contains the last
// level also writes the frame's keypoint count (K3 must have finished for all levels by then).
void pg_launch_describe_levels(const PgPlan& P, int nframes, pgorb_keypoint* d_kps, uint8_t* d_desc,
                               int cap_per_frame, int32_t* d_n, int levelBeg, int levelEnd, hipStream_t s)
{
    static const PgGauss7 G = pg_gauss7();
    static bool tabReady[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !tabReady[dev]) {
        const int K[7] = {G.k0, G.k1, G.k2, G.k3, G.k2, G.k1, G.k0};
        uint32_t tab[3 * 64 * 4] = {};
        for (int cb = 0; cb < 3; cb++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 16; j++) {
                    const int k = 16 * (l >> 4) + j, n = 16 * cb + (l & 15), t = k - n;
                    const uint32_t v = (t >= 0 && t <= 6 && n < DH_N) ? (uint32_t)K[t] : 0u;
                    tab[(cb * 64 + l) * 4 + j / 4] |= v << (8 * (j % 4));
                }
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(pg_blur_btab), tab, sizeof(tab), 0, hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);                           // tab is a stack array
        tabReady[dev] = true;
    }
    const int slotBeg = (int)P.lvl[levelBeg].selOff;
    const int slotEnd = (levelEnd < P.nlevels) ? (int)P.lvl[levelEnd].selOff : P.selTotal;
    dim3 grid((slotEnd - slotBeg + 7) & ~7, nframes), block(64);
    hipLaunchKernelGGL(k_describe, grid, block, 0, s, P, G, d_kps, d_desc, cap_per_frame, d_n, slotBeg, slotEnd,
                       levelEnd == P.nlevels ? 1 : 0, 0);
}

// ---- exhaustive check of the sin/cos contract (tests/test_gpu_parity.py, SURVEY.md hard part 4) --------
// Block b folds pg_sincos_f of the `count` floats whose bit patterns start at first_bits + b * count into one
// 64-bit checksum: sum over i of (sin bits * 0x9E3779B1 + cos bits) * (2 i + 1)  (mod 2^64) -- the oracle's
// orc_sincos_checksum computes the same sum with orc_sincos_f.
__global__ __launch_bounds__(256) void k_sincos_checksum(uint32_t first_bits, uint32_t count, unsigned long long* out)
{
    const uint32_t base = first_bits + blockIdx.x * count;
    unsigned long long acc = 0;
    for (uint32_t i = threadIdx.x; i < count; i += 256) {
        float s, c;
        pg_sincos_f(__uint_as_float(base + i), &s, &c);
        acc += ((unsigned long long)__float_as_uint(s) * 0x9E3779B1ull + __float_as_uint(c)) * (2ull * i + 1ull);
    }
    __shared__ unsigned long long red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

extern "C" int pgorb_debug_sincos_checksum(uint32_t first_bits, uint32_t count, int nblocks, unsigned long long* out)
{
    if (!out || nblocks < 1 || count < 1) return PGORB_E_ARG;
    unsigned long long* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(unsigned long long) * (size_t)nblocks) != hipSuccess) return PGORB_E_HIP;
    hipLaunchKernelGGL(k_sincos_checksum, dim3(nblocks), dim3(256), 0, 0, first_bits, count, d);
    const bool ok = hipMemcpy(out, d, sizeof(unsigned long long) * (size_t)nblocks, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? 0 : PGORB_E_HIP;
}

// match.hip -- K7: 256-bit Hamming distance kernels (gfx950).
//
// Restates ORBmatcher::DescriptorDistance (thirdparty/orb-slam2/src/ORBmatcher.cc:1651-1667,
// identical to DBoW2 FORB::distance, thirdparty/DBoW2/DBoW2/FORB.cpp:81-101): the sum of set
// bits of the XOR of two 32-byte descriptors.  The reference's SWAR bit-hack is the integer
// popcount.  The all-pairs matrix and the best/second-best scan are the superset of the
// reference matchers' candidate loops (bestDist / bestDist2 with strict '<',
// ORBmatcher.cc:438-459).  Two implementations: v_bcnt_u32_b32 on eight 32-bit words (the
// distance matrix, and best-2 for frames of 8192+ descriptors: each lane keeps ONE query
// descriptor in 8 VGPRs and streams the train descriptors through LDS as wave-uniform
// 128-bit reads), and the i8 MFMA formulation below for the batch matcher.
//
// Bytes: 32 B per descriptor read once per query block; bound by popcount issue / the matrix
// pipe, not by HBM.
#include "pgorb_internal.h"

#define MT_T 64
#define MT_TILE 256           // train descriptors staged per LDS tile (8 KiB)

__device__ __forceinline__ int pg_hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ void stage_tile(uint4* tile, const uint8_t* b, int j0, int nb, int tid, int nthreads)
{
    const int cnt = min(MT_TILE, nb - j0);
    const uint4* src = reinterpret_cast<const uint4*>(b + (int64_t)j0 * 32);
    for (int i = tid; i < 2 * cnt; i += nthreads) tile[i] = src[i];
}

__global__ __launch_bounds__(MT_T) void k_hamming_matrix(const uint8_t* __restrict__ a, int na,
                                                          const uint8_t* __restrict__ b, int nb,
                                                          uint16_t* __restrict__ out)
{
    __shared__ uint4 tile[2 * MT_TILE];
    const int i = blockIdx.x * MT_T + threadIdx.x;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (i < na) {
        q0 = reinterpret_cast<const uint4*>(a + (int64_t)i * 32)[0];
        q1 = reinterpret_cast<const uint4*>(a + (int64_t)i * 32)[1];
    }
    for (int j0 = blockIdx.y * MT_TILE; j0 < nb; j0 += gridDim.y * MT_TILE) {
        __syncthreads();
        stage_tile(tile, b, j0, nb, threadIdx.x, MT_T);
        __syncthreads();
        const int cnt = min(MT_TILE, nb - j0);
        if (i < na)
            for (int j = 0; j < cnt; j++)
                out[(int64_t)i * nb + j0 + j] = (uint16_t)pg_hamming256(q0, q1, tile[2 * j], tile[2 * j + 1]);
    }
}

// ---- best / second-best over all train descriptors: the all-pairs scan on the matrix cores ----
//
// With bits mapped to +1 / -1 the dot product of two 256-element vectors is 256 - 2 * Hamming, so
// a 16 x 16 tile of distances is TWO v_mfma_scale_f32_16x16x128_f8f6f4 on fp4 (e2m1) operands
// instead of 16 * 16 * (8 xor + 8 bcnt); fp4 is the format with the highest MFMA rate on gfx950
// and +-1 is all a bit needs.  The block scales (E8M0, 2^6 on either side) multiply the products
// by 4096 and the first MFMA of a tile takes its C operand from a per-lane constant, so the f32
// accumulator leaves the matrix pipe as the finished sort key
//     key = (256 - distance) * 8192 + (8191 - train index)            (train index < 8192)
// -- an integer below 2^22, exact in f32 -- and what stays on the VALU is a median / maximum pair per
// distance (three instructions per two distances, mx_update2), which keep the two LARGEST keys (k1 >= k2) of every (row, column class): larger key =
// smaller distance, then smaller index -- the reference's strict-'<' scan with bestDist /
// bestDist2 (ORBmatcher.cc:438-459), "first minimum wins".  (dot + 256 is even, which is where
// the 13th index bit comes from.)  tools/ubench/mfma_fp4_dot.hip checks the instruction's
// semantics (D = C + 4096 * dot exactly; row 4 * (l >> 4) + r, column l & 15).
//
// Operand layout ("x16 block", 2 KiB per 16 descriptors): [k-step s][lane l][16 B]; the entry of
// lane l in k-step s holds bits 128 s + 32 (l >> 4) .. + 31 of descriptor l & 15, one fp4 per bit
// (0x2 = +1.0, 0xA = -1.0), bit e in nibble e.  Rows/columns are l & 15 for A and B alike and both
// sides use the same K order, which is all the dot product needs.  Queries are expanded in registers.  Train descriptors
// come into LDS as they are, 128 at a time by LDS-DMA (double buffered), and the workgroup -- 16 waves, 1 024 queries --
// expands the tile into this layout in LDS once for all its waves (round 3; rounds 1-2 expanded every train frame into a
// scratch slab in HBM first, k_expand_trains: kept as PGORB_MATCH_MODE=0).
typedef int pg_v4i __attribute__((ext_vector_type(4)));
typedef int pg_v8i __attribute__((ext_vector_type(8)));
typedef float pg_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* pg_gptr_t;
typedef __attribute__((address_space(3))) void* pg_lptr_t;

#define MX_BLOCK_BYTES 2048
#define MX_TILE_BLOCKS 8          // train blocks staged per LDS tile (128 descriptors, 16 KiB)
#ifndef MX_TILE_BLOCKS_WIDE
#define MX_TILE_BLOCKS_WIDE 16    // ... per tile of the 16-wave form (256 descriptors: 0.0787 -> 0.0772 ms against 8; 24: 0.0770)
#endif
#define MX_WAVES 4                // 64 queries per wave, 256 per workgroup
#define MX_IDX_BITS 13
#define MX_IDX_MASK ((1 << MX_IDX_BITS) - 1)
#define MX_MAX_TRAIN (1 << MX_IDX_BITS)
#define MX_SCALE 0x85858585       // E8M0 133 = 2^6 per side: products x 4096

// 8 bits -> 8 fp4 values, bit e in nibble e: +1.0 (0x2) where set, -1.0 (0xA) where clear
__device__ __forceinline__ uint32_t pg_fp4x8(uint32_t b)
{
    uint32_t x = (b | (b << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;                            // nibble e = bit e
    return (x << 3) ^ 0xAAAAAAAAu;
}

__device__ __forceinline__ pg_v4i pg_fp4x32(uint32_t bits32)
{
    pg_v4i v;
    v.x = (int)pg_fp4x8(bits32 & 0xFFu); v.y = (int)pg_fp4x8((bits32 >> 8) & 0xFFu);
    v.z = (int)pg_fp4x8((bits32 >> 16) & 0xFFu); v.w = (int)pg_fp4x8(bits32 >> 24);
    return v;
}

// grid (blocks of 16 train descriptors, pairs); 128 threads: thread t writes entry (s = t / 64, lane = t % 64)
__global__ __launch_bounds__(128) void k_expand_trains(const uint8_t* __restrict__ tdesc, const int32_t* __restrict__ n,
                                                        int cap, const int32_t* __restrict__ pt, int nb_single,
                                                        int blocksPerPair, uint8_t* __restrict__ xt)
{
    const int p = blockIdx.y, tb = blockIdx.x;
    const int ft = pt ? pt[p] : 0;
    const int nb = pt ? min(n[ft], cap) : nb_single;
    if (16 * tb >= nb) return;                                   // never read
    const int s = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int idx = 16 * tb + (l & 15);
    pg_v4i v = {0, 0, 0, 0};                                     // fp4 0.0: padding columns (masked anyway)
    if (idx < nb) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(tdesc + ((int64_t)ft * cap + idx) * 32);
        v = pg_fp4x32(src[4 * s + (l >> 4)]);
    }
    *reinterpret_cast<pg_v4i*>(xt + ((int64_t)p * blocksPerPair + tb) * MX_BLOCK_BYTES + threadIdx.x * 16) = v;
}

// Keys are positive integers held as f32 values; -1 is "none".  v_med3_f32 is a compiler-known
// instruction, so the hazard recogniser places the wait states between the MFMA that writes a
// VGPR and its first VALU read (match.hip is built with MFMA results in VGPRs, see the Makefile).
__device__ __forceinline__ float pg_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// keys of this wave's 64 queries (A) against the 16 trains of one staged x16 block (B, already
// in registers); kq = the per-lane key offset 256 * 4096 + (8191 - column), in all four registers
__device__ __forceinline__ void mx_block(const pg_v4i (&B)[2], const pg_v4i (&A)[4][2], const pg_v4f kq, pg_v4f (&acc)[4])
{
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const pg_v8i b8 = {B[s].x, B[s].y, B[s].z, B[s].w, 0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const pg_v8i a8 = {A[a][s].x, A[a][s].y, A[a][s].z, A[a][s].w, 0, 0, 0, 0};
            acc[a] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, s == 0 ? kq : acc[a], 4, 4, 0, MX_SCALE, 0, MX_SCALE);
        }
    }
}

__device__ __forceinline__ void mx_load(const uint8_t* blockLane, pg_v4i (&B)[2])
{
#pragma unroll
    for (int s = 0; s < 2; s++) B[s] = *reinterpret_cast<const pg_v4i*>(blockLane + s * 1024);
}

template <bool MASK>
__device__ __forceinline__ void mx_update(const pg_v4f (&acc)[4], bool valid, float (&k1)[4][4], float (&k2)[4][4])
{
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float key = acc[a][r];
            if (MASK) key = valid ? key : -1.f;
            k2[a][r] = pg_med3(k1[a][r], k2[a][r], key);
            k1[a][r] = pg_med3(k1[a][r], key, __builtin_inff());     // max(k1, key): the median with +inf
        }
}

// the keys of TWO blocks at once: with x, y the new keys of a (row, column class), the largest of {k1, k2, x, y} is max3(k1, x, y) and
// the second largest max(med3(k1, x, y), k2) (k2 <= k1) -- three instructions for two distances instead of four
__device__ __forceinline__ void mx_update2(const pg_v4f (&accA)[4], const pg_v4f (&accB)[4], float (&k1)[4][4], float (&k2)[4][4])
{
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float x = accA[a][r], y = accB[a][r];
            k2[a][r] = __builtin_fmaxf(pg_med3(k1[a][r], x, y), k2[a][r]);
            k1[a][r] = __builtin_fmaxf(__builtin_fmaxf(k1[a][r], x), y);
        }
}

// grid (pairs padded to a multiple of 8, ceil(cap / 256)): the PAIR is the fast grid index, so the workgroup's
// linear id mod 8 -- the XCD it is dispatched to -- is pair mod 8 and all query blocks of one pair share one XCD's
// L2: the pair's expanded trains (128 B per descriptor) come from HBM once instead of once per query block
// (profiles/r01_i_pmc.txt: 264 MB per step fetched for 18 MB of descriptors).  nb < 8192
// WAVES waves of 64 queries per workgroup.  RAW: `xt` is the train frame set itself (32-byte descriptors): a tile's 128
// descriptors come into LDS as they are (4 KiB by LDS-DMA, double buffered) and the WORKGROUP expands them to the fp4 operand
// layout in LDS -- each thread a share of the tile's 1 024 sixteen-byte entries -- before its waves consume them; no expanded
// copy in HBM.  !RAW: `xt` is k_expand_trains' slab, tiles arrive expanded (16 KiB each, double buffered).
template <int WAVES, bool RAW>
__global__ __launch_bounds__(64 * WAVES) void k_match_mfma(const uint8_t* __restrict__ qdesc, const uint8_t* __restrict__ xt,
                                                           const int32_t* __restrict__ n, int cap,
                                                           const int32_t* __restrict__ pq, const int32_t* __restrict__ pt,
                                                           int na_single, int nb_single, int blocksPerPair, int npairs,
                                                           int32_t* best_idx, uint16_t* best, uint16_t* second)
{
    // blocks per staged tile: 8 (128 descriptors); the 16-wave raw form has the LDS for more and fewer, longer tiles mean fewer barriers
    constexpr int TB = (RAW && WAVES == 16) ? MX_TILE_BLOCKS_WIDE : MX_TILE_BLOCKS;
    constexpr int TILE_BYTES = TB * MX_BLOCK_BYTES, RAW_BYTES = TB * 16 * 32;
    // !RAW: two expanded tiles + read-ahead slack.  RAW: one expanded tile + slack, then two raw tiles
    constexpr int EXP_BYTES = (RAW ? 1 : 2) * TILE_BYTES + MX_BLOCK_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t tile[EXP_BYTES + (RAW ? 2 * RAW_BYTES : 0)];
    const int p = blockIdx.x, qblk = blockIdx.y;
    if (p >= npairs) return;
    const int fq = pq ? pq[p] : 0, ft = pt ? pt[p] : 0;
    const int na = pq ? min(n[fq], cap) : na_single, nb = pt ? min(n[ft], cap) : nb_single;
    if (qblk * 64 * WAVES >= na) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qbase = qblk * 64 * WAVES + wv * 64;
    const bool active = qbase < na;                              // wave-uniform
    const uint8_t* qd = qdesc + (int64_t)fq * cap * 32;
    const int64_t o = (int64_t)p * cap;

    // queries of this wave as MFMA A operands: A[a][s] = rows 16a .. 16a+15, k-step s (filled below, behind the first tile's request)
    pg_v4i A[4][2];
    float k1[4][4], k2[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) { k1[a][r] = -1.f; k2[a][r] = -1.f; }      // -1 = none (valid keys are >= 0)

    const int nblocks = (nb + 15) >> 4;
    const uint8_t* xp = RAW ? xt + (int64_t)ft * cap * 32 : xt + (int64_t)p * blocksPerPair * MX_BLOCK_BYTES;
    uint8_t* rawBase = tile + EXP_BYTES;                         // (RAW)
    // double-buffered tiles: the LDS-DMA of tile t+1 is in flight while tile t is consumed
    auto stage = [&](int t0, int buf) {
        if (RAW) {
            // 16 B per lane: chunk c = 64 wv + lane is half (c & 1) of descriptor t0 * 16 + (c >> 1); descriptors past nb are
            // not read (their columns are masked by the update, their LDS bytes are whatever was there)
            for (int c0 = 64 * wv; c0 < RAW_BYTES / 16; c0 += 64 * WAVES) {
                const int c = c0 + lane, d = t0 * 16 + (c >> 1);
                if (d < nb)
                    __builtin_amdgcn_global_load_lds((pg_gptr_t)(xp + (int64_t)d * 32 + (c & 1) * 16),
                                                     (pg_lptr_t)(rawBase + buf * RAW_BYTES + c0 * 16), 16, 0, 0);
            }
        } else {
            const int cnt = min(TB, nblocks - t0);
            // 16 B per lane, wave w fills KiB w, w + WAVES, ... of the tile
            for (int kb = wv; kb < cnt * (MX_BLOCK_BYTES / 1024); kb += WAVES)
                __builtin_amdgcn_global_load_lds((pg_gptr_t)(xp + (int64_t)t0 * MX_BLOCK_BYTES + kb * 1024 + lane * 16),
                                                 (pg_lptr_t)(tile + buf * TILE_BYTES + kb * 1024), 16, 0, 0);
        }
    };
    // one staged tile (blocks t0 .. t0 + cnt - 1, operands at `bl` for this lane) against this wave's 64 queries
    auto compute_tile = [&](const uint8_t* bl, int t0, int cnt) {
        float kc = (float)((256 << (MX_IDX_BITS - 1)) + MX_IDX_MASK - (t0 * 16 + (lane & 15)));
        // blocks that lie entirely below nb take the unmasked update; at most one block per
        // pair is partial (kept out of the hot loop: a select inside it costs 32 register copies)
        const int cntFull = min(cnt, (nb >> 4) - t0);
        // ping-pong B operands: block blk+1 is fetched from LDS while block blk is in the matrix
        // pipe (reads past the last block of the tile stay inside the double buffer)
        pg_v4i B0[2], B1[2];
        pg_v4f acc[4];
        mx_load(bl, B0);
        int blk = 0;
        for (; blk + 2 <= cntFull; blk += 2, kc -= 32.f) {
            mx_load(bl + (blk + 1) * MX_BLOCK_BYTES, B1);
            mx_block(B0, A, pg_v4f{kc, kc, kc, kc}, acc);
            pg_v4f accB[4];
            mx_block(B1, A, pg_v4f{kc - 16.f, kc - 16.f, kc - 16.f, kc - 16.f}, accB);
            mx_load(bl + (blk + 2) * MX_BLOCK_BYTES, B0);
            mx_update2(acc, accB, k1, k2);
        }
        if (blk < cntFull) {                                     // odd count: one more full block, in B0
            mx_load(bl + (blk + 1) * MX_BLOCK_BYTES, B1);
            mx_block(B0, A, pg_v4f{kc, kc, kc, kc}, acc);
            mx_update<false>(acc, true, k1, k2);
            blk++; kc -= 16.f;
            if (blk < cnt) {
                mx_block(B1, A, pg_v4f{kc, kc, kc, kc}, acc);
                mx_update<true>(acc, (t0 + blk) * 16 + (lane & 15) < nb, k1, k2);
            }
        } else if (blk < cnt) {                                  // the partial block is in B0
            mx_block(B0, A, pg_v4f{kc, kc, kc, kc}, acc);
            mx_update<true>(acc, (t0 + blk) * 16 + (lane & 15) < nb, k1, k2);
        }
    };
    if (nblocks > 0) stage(0, 0);
    // (the query loads and their expansion run while the first train tile is on its way)
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int q = qbase + 16 * a + (lane & 15);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(qd + (int64_t)q * 32) + (lane >> 4);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            A[a][s] = pg_v4i{0, 0, 0, 0};
            if (q < na) A[a][s] = pg_fp4x32(src[4 * s]);
        }
    }
    int buf = 0;
    for (int t0 = 0; t0 < nblocks; t0 += TB, buf ^= 1) {
        const int cnt = min(TB, nblocks - t0);
        __builtin_amdgcn_s_waitcnt(0);                           // this wave's share of tile t0 has landed
        __syncthreads();                                         // ... everybody's; and buffer buf^1 (RAW: the expanded tile) has been consumed
        if (!RAW && t0 + TB < nblocks) stage(t0 + TB, buf ^ 1);
        if (RAW) {
            // entry e = (block e >> 7, k-step (e >> 6) & 1, lane e & 63): bits 128 s + 32 (l >> 4) .. of descriptor l & 15
            const uint32_t* raw32 = reinterpret_cast<const uint32_t*>(rawBase + buf * RAW_BYTES);
            for (int e = threadIdx.x; e < TILE_BYTES / 16; e += 64 * WAVES) {
                const int l = e & 63, s = (e >> 6) & 1, blk = e >> 7;
                *reinterpret_cast<pg_v4i*>(tile + e * 16) = pg_fp4x32(raw32[(blk * 16 + (l & 15)) * 8 + 4 * s + (l >> 4)]);
            }
            __syncthreads();                                     // the expanded tile is complete
            // (the next raw tile is requested only now: with an LDS-DMA outstanding the compiler holds every ds_write behind
            //  vmcnt(0) -- it cannot tell the two LDS targets apart -- and the expansion would wait for the load it should hide)
            if (t0 + TB < nblocks) stage(t0 + TB, buf ^ 1);
        }
        if (!active) continue;
        compute_tile(tile + (RAW ? 0 : buf * TILE_BYTES) + lane * 16, t0, cnt);
    }
    if (!active) return;
    // merge the 16 column classes of a row (lanes with equal lane >> 4)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int m1 = (int)k1[a][r], m2 = (int)k2[a][r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                const int o1 = __shfl_xor(m1, m), o2 = __shfl_xor(m2, m);
                const int hi = max(m1, o1);
                m2 = max(min(m1, o1), max(m2, o2));
                m1 = hi;
            }
            const int q = qbase + 16 * a + 4 * (lane >> 4) + r;
            if ((lane & 15) == 0 && q < na) {
                best_idx[o + q] = m1 >= 0 ? (int32_t)(MX_IDX_MASK - (m1 & MX_IDX_MASK)) : -1;
                best[o + q] = m1 >= 0 ? (uint16_t)(256 - (m1 >> MX_IDX_BITS)) : (uint16_t)65535;
                second[o + q] = m2 >= 0 ? (uint16_t)(256 - (m2 >> MX_IDX_BITS)) : (uint16_t)65535;
            }
        }
}

// ---- popcount path: frames with 8192 or more descriptors (the MFMA key has 13 index bits) ----
// best / second-best over all train descriptors (nb < 2^20) for 64 queries per workgroup.  The workgroup
// is MT_WAVES waves: wave w scans the w-th contiguous slice of the train set for the SAME 64
// queries (its own LDS tile), then wave 0 merges the partial results in slice order, which
// preserves "first minimum wins" (strict '<', ORBmatcher.cc:443-458).
#define MT_WAVES 4
#define MT_SLICE_TILE 128     // train descriptors staged per wave and LDS tile (4 KiB)

// key = distance << 20 | train index: the two smallest keys of a query are its best and second
// best in (distance, first index) order, i.e. exactly the reference's strict-'<' scan; 4 VALU per
// candidate instead of a compare/select chain.
#define MT_IDX_BITS 20

__device__ __forceinline__ void best2_scan(const uint8_t* a, int na, const uint8_t* b, int nb,
                                           int32_t* best_idx, uint16_t* best, uint16_t* second,
                                           uint4* tiles, unsigned* part)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * MT_T + lane;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (i < na) {
        q0 = reinterpret_cast<const uint4*>(a + (int64_t)i * 32)[0];
        q1 = reinterpret_cast<const uint4*>(a + (int64_t)i * 32)[1];
    }
    const int per = (nb + MT_WAVES - 1) / MT_WAVES;
    const int jb = wv * per, je = min(jb + per, nb);
    uint4* tile = tiles + wv * 2 * MT_SLICE_TILE;
    unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;                // two smallest keys of this slice
    for (int j0 = jb; j0 < je; j0 += MT_SLICE_TILE) {          // waves run independently: wave-local sync only
        const int cnt = min(MT_SLICE_TILE, je - j0);
        const uint4* src = reinterpret_cast<const uint4*>(b + (int64_t)j0 * 32);
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < 2 * cnt; k += 64) tile[k] = src[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int j = 0; j < cnt; j++) {
            const unsigned d = (unsigned)pg_hamming256(q0, q1, tile[2 * j], tile[2 * j + 1]);
            const unsigned key = (d << MT_IDX_BITS) | (unsigned)(j0 + j);
            k2 = min(k2, max(k1, key));
            k1 = min(k1, key);
        }
    }
    part[(wv * 64 + lane) * 2 + 0] = k1;
    part[(wv * 64 + lane) * 2 + 1] = k2;
    __syncthreads();
    if (wv == 0 && i < na) {
        unsigned m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < 2 * MT_WAVES; w++) {
            const unsigned key = part[(w >> 1) * 128 + lane * 2 + (w & 1)];
            m2 = min(m2, max(m1, key));
            m1 = min(m1, key);
        }
        best_idx[i] = (m1 == 0xFFFFFFFFu) ? -1 : (int32_t)(m1 & ((1u << MT_IDX_BITS) - 1));
        best[i] = (m1 == 0xFFFFFFFFu) ? (uint16_t)65535 : (uint16_t)(m1 >> MT_IDX_BITS);
        second[i] = (m2 == 0xFFFFFFFFu) ? (uint16_t)65535 : (uint16_t)(m2 >> MT_IDX_BITS);
    }
}

__global__ __launch_bounds__(MT_T * MT_WAVES) void k_hamming_best2(const uint8_t* __restrict__ a, int na,
                                                         const uint8_t* __restrict__ b, int nb,
                                                         int32_t* best_idx, uint16_t* best, uint16_t* second)
{
    __shared__ uint4 tiles[MT_WAVES * 2 * MT_SLICE_TILE];
    __shared__ unsigned part[MT_WAVES * 64 * 2];
    best2_scan(a, na, b, nb, best_idx, best, second, tiles, part);
}

__global__ __launch_bounds__(MT_T * MT_WAVES) void k_match_batch(const uint8_t* __restrict__ desc,
                                                       const int32_t* __restrict__ n, int cap,
                                                       const int32_t* __restrict__ pq,
                                                       const int32_t* __restrict__ pt,
                                                       int32_t* best_idx, uint16_t* best, uint16_t* second)
{
    __shared__ uint4 tiles[MT_WAVES * 2 * MT_SLICE_TILE];
    __shared__ unsigned part[MT_WAVES * 64 * 2];
    const int p = blockIdx.y;
    const int fq = pq[p], ft = pt[p];
    const int na = min(n[fq], cap), nb = min(n[ft], cap);
    if ((int)blockIdx.x * MT_T >= na) return;
    const int64_t o = (int64_t)p * cap;
    best2_scan(desc + (int64_t)fq * cap * 32, na, desc + (int64_t)ft * cap * 32, nb,
               best_idx + o, best + o, second + o, tiles, part);
}

void pg_launch_hamming_matrix(const uint8_t* d_a, int na, const uint8_t* d_b, int nb,
                              uint16_t* d_out, hipStream_t s)
{
    if (na <= 0 || nb <= 0) return;
    int gy = (nb + MT_TILE - 1) / MT_TILE;
    if (gy > 64) gy = 64;
    dim3 grid((na + MT_T - 1) / MT_T, gy), block(MT_T);
    hipLaunchKernelGGL(k_hamming_matrix, grid, block, 0, s, d_a, na, d_b, nb, d_out);
}

// PGORB_MATCH_POPCOUNT=1 (or pgorb_set_option "matcher" 1) forces the v_bcnt kernels for every size (the variant
// BASELINE.json's north star describes), so both matchers can be timed on the same frames; results are identical.
// how the train descriptors reach the matrix cores: 2 = taken as they are and expanded in LDS by workgroups of 16 waves (1 024 queries
// share one expansion of a tile: 0.077 ms per 127 pairs of 2 000, no expanded copy in HBM); 1 = the same with 4-wave workgroups
// (0.099: the expansion is repeated per 256 queries); 0 = expanded once per pair into a scratch slab by k_expand_trains (0.098,
// 120 MB of traffic per step for 18.5 MB of descriptors).  Default: 2 when that makes at least 192 workgroups (a 16-wave workgroup
// takes a whole CU's matrix time: 31 pairs of 4 006 are 124 of them and ran 0.128 ms against 0.10), else 1.
// PGORB_MATCH_MODE = 0 | 1 | 2, or pgorb_set_option("match_mode", ...) (-1 = by grid size again), forces one (measurement switch).
// Both settings belong to a CONTEXT (PgMatchOpts, round 4; they were file-scope statics): the environment only seeds pgorb_create.
PgMatchOpts pg_match_default_opts()
{
    PgMatchOpts o;
    o.popcount = getenv("PGORB_MATCH_POPCOUNT") != nullptr ? 1 : 0;
    const char* e = getenv("PGORB_MATCH_MODE");
    const int m = e ? atoi(e) : -1;
    o.mode = (m >= 0 && m <= 2) ? m : -1;
    return o;
}
static int mx_mode(const PgMatchOpts& o, int npairs, int nq)
{
    if (o.mode >= 0) return o.mode;
    return (long long)npairs * ((nq + 1023) / 1024) >= 192 ? 2 : 1;
}
bool pg_match_uses_popcount(const PgMatchOpts& o, int cap_per_frame) { return cap_per_frame >= MX_MAX_TRAIN || o.popcount; }

size_t pg_match_scratch_bytes(const PgMatchOpts& o, int nb_max, int npairs)
{
    if (o.mode != 0 || pg_match_uses_popcount(o, nb_max)) return 0;      // (only the slab form needs scratch)
    return (size_t)npairs * (size_t)((nb_max + 15) / 16) * MX_BLOCK_BYTES;
}

// single pair: a (na descriptors) against b (nb descriptors); scratch >= pg_match_scratch_bytes(nb, 1)
void pg_launch_best2(const PgMatchOpts& o, const uint8_t* d_a, int na, const uint8_t* d_b, int nb, uint8_t* d_scratch,
                     int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second, hipStream_t s)
{
    if (na <= 0) return;
    if (pg_match_uses_popcount(o, nb)) {
        hipLaunchKernelGGL(k_hamming_best2, dim3((na + MT_T - 1) / MT_T), dim3(MT_T * MT_WAVES), 0, s, d_a, na, d_b, nb,
                           d_best_idx, d_best, d_second);
        return;
    }
    const int bpp = (nb + 15) / 16;
    const int mode = mx_mode(o, 1, na);
    if (mode == 0) {
        if (bpp > 0)
            hipLaunchKernelGGL(k_expand_trains, dim3(bpp, 1), dim3(128), 0, s, d_b, nullptr, nb, nullptr, nb, bpp, d_scratch);
        hipLaunchKernelGGL((k_match_mfma<4, false>), dim3(1, (na + 255) / 256), dim3(256), 0, s,
                           d_a, d_scratch, nullptr, na, nullptr, nullptr, na, nb, bpp, 1, d_best_idx, d_best, d_second);
    } else if (mode == 1) {
        hipLaunchKernelGGL((k_match_mfma<4, true>), dim3(1, (na + 255) / 256), dim3(256), 0, s,
                           d_a, d_b, nullptr, nb, nullptr, nullptr, na, nb, bpp, 1, d_best_idx, d_best, d_second);
    } else {
        hipLaunchKernelGGL((k_match_mfma<16, true>), dim3(1, (na + 1023) / 1024), dim3(1024), 0, s,
                           d_a, d_b, nullptr, nb, nullptr, nullptr, na, nb, bpp, 1, d_best_idx, d_best, d_second);
    }
}

void pg_launch_match_batch(const PgMatchOpts& o, const uint8_t* d_desc, const int32_t* d_n, int cap_per_frame,
                           const int32_t* d_pq, const int32_t* d_pt, int npairs, uint8_t* d_scratch,
                           int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second, hipStream_t s)
{
    if (npairs <= 0) return;
    if (pg_match_uses_popcount(o, cap_per_frame)) {
        hipLaunchKernelGGL(k_match_batch, dim3((cap_per_frame + MT_T - 1) / MT_T, npairs), dim3(MT_T * MT_WAVES), 0, s,
                           d_desc, d_n, cap_per_frame, d_pq, d_pt, d_best_idx, d_best, d_second);
        return;
    }
    const int bpp = (cap_per_frame + 15) / 16;
    const int mode = mx_mode(o, npairs, cap_per_frame);
    const dim3 grid4((npairs + 7) & ~7, (cap_per_frame + 255) / 256), grid16((npairs + 7) & ~7, (cap_per_frame + 1023) / 1024);
    if (mode == 0) {
        hipLaunchKernelGGL(k_expand_trains, dim3(bpp, npairs), dim3(128), 0, s, d_desc, d_n, cap_per_frame, d_pt, 0, bpp, d_scratch);
        hipLaunchKernelGGL((k_match_mfma<4, false>), grid4, dim3(256), 0, s,
                           d_desc, d_scratch, d_n, cap_per_frame, d_pq, d_pt, 0, 0, bpp, npairs, d_best_idx, d_best, d_second);
    } else if (mode == 1) {
        hipLaunchKernelGGL((k_match_mfma<4, true>), grid4, dim3(256), 0, s,
                           d_desc, d_desc, d_n, cap_per_frame, d_pq, d_pt, 0, 0, bpp, npairs, d_best_idx, d_best, d_second);
    } else {
        hipLaunchKernelGGL((k_match_mfma<16, true>), grid16, dim3(1024), 0, s,
                           d_desc, d_desc, d_n, cap_per_frame, d_pq, d_pt, 0, 0, bpp, npairs, d_best_idx, d_best, d_second);
    }
}

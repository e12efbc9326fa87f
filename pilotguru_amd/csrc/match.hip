// match.hip -- K7: 256-bit Hamming distance kernels (gfx950).
//
// Restates ORBmatcher::DescriptorDistance (thirdparty/orb-slam2/src/ORBmatcher.cc:1651-1667,
// identical to DBoW2 FORB::distance, thirdparty/DBoW2/DBoW2/FORB.cpp:81-101): the sum of set
// bits of the XOR of two 32-byte descriptors.  The reference's SWAR bit-hack is the integer
// popcount; here it is v_bcnt_u32_b32 on eight 32-bit words.  The all-pairs matrix and the
// best/second-best scan are the superset of the reference matchers' candidate loops
// (bestDist / bestDist2 with strict '<', ORBmatcher.cc:438-459): each lane keeps ONE query
// descriptor in 8 VGPRs and streams the train descriptors through LDS as wave-uniform
// (broadcast, conflict-free) 128-bit reads.
//
// Bytes: 32 B per descriptor read once per 64-query block; popcount-bound, not HBM-bound.
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

void pg_launch_best2(const uint8_t* d_a, int na, const uint8_t* d_b, int nb,
                     int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second, hipStream_t s)
{
    if (na <= 0) return;
    dim3 grid((na + MT_T - 1) / MT_T), block(MT_T * MT_WAVES);
    hipLaunchKernelGGL(k_hamming_best2, grid, block, 0, s, d_a, na, d_b, nb, d_best_idx, d_best, d_second);
}

void pg_launch_match_batch(const uint8_t* d_desc, const int32_t* d_n, int cap_per_frame,
                           const int32_t* d_pq, const int32_t* d_pt, int npairs,
                           int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second, hipStream_t s)
{
    if (npairs <= 0) return;
    dim3 grid((cap_per_frame + MT_T - 1) / MT_T, npairs), block(MT_T * MT_WAVES);
    hipLaunchKernelGGL(k_match_batch, grid, block, 0, s, d_desc, d_n, cap_per_frame, d_pq, d_pt,
                       d_best_idx, d_best, d_second);
}

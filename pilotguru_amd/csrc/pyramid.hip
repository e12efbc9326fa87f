// pyramid.hip -- K1: bilinear pyramid chain for gfx950.
//
// Restates ORBextractor::ComputePyramid (thirdparty/orb-slam2/src/ORBextractor.cc:
// 1106-1131): level l = cv::resize(level l-1, INTER_LINEAR) (:1119), no pre-blur.  The
// 11-bit fixed-point arithmetic is OpenCV 2.4's; coefficient
// tables are built once per frame size on the host (api.hip: build_resize_tables) with the
// same float/double sequence as cv::resize.  The 19-px reflect border the reference adds
// (:1121,:1126) is never read on the monocular path, so planes are stored unpadded.
//
// HBM-bound u8 streaming kernels, fastest first (pg_launch_pyramid_level picks):
//   k_pyr_resize_rows4_lds   256 x 32 destination tile per workgroup, source rectangle staged through
//                            LDS by LDS-DMA; lane = 4 columns x 4 rows x 2 groups (scale <= ~1.4)
//   k_pyr_resize_rows4       the same arithmetic with per-lane unaligned 8-byte global windows
//   k_pyr_resize_quads / k_pyr_resize_bilinear_u8   generic scale factors, 4 pixels per lane from
//                            two source rows read as bytes
// Algorithmic bytes per launch: w_{l-1}*h_{l-1} read + w_l*h_l written, per frame.
#include "pgorb_internal.h"

__global__ __launch_bounds__(256) void k_copy_level0(const uint8_t* __restrict__ src, int stride,
                                                      int64_t sfstride, uint8_t* __restrict__ dst,
                                                      int pitch, int64_t dfstride, int w, int h)
{
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= w || y >= h) return;
    const uint8_t* s = src + (int64_t)blockIdx.z * sfstride + (int64_t)y * stride + x4;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x4 + k < w) v |= (uint32_t)s[k] << (8 * k);
    *reinterpret_cast<uint32_t*>(dst + (int64_t)blockIdx.z * dfstride + (int64_t)y * pitch + x4) = v;
}

// cvtColor RGB/BGR(A) -> GRAY, 8U fixed point (OpenCV 2.4 color.cpp RGB2Gray<uchar>: yuv_shift 14,
// coefficients R 4899, G 9617, B 1868), 4 pixels per lane, written into the level-0 plane.
__global__ __launch_bounds__(256) void k_color_to_gray(const uint8_t* __restrict__ src, int stride,
                                                        int64_t sfstride, int cn, int rIdx,
                                                        uint8_t* __restrict__ dst, int pitch,
                                                        int64_t dfstride, int w, int h)
{
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= w || y >= h) return;
    const uint8_t* s = src + (int64_t)blockIdx.z * sfstride + (int64_t)y * stride + (int64_t)x4 * cn;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x4 + k < w) {
            const uint8_t* px = s + k * cn;
            const int R = px[rIdx], G = px[1], B = px[2 - rIdx];
            v |= (uint32_t)((R * 4899 + G * 9617 + B * 1868 + 8192) >> 14) << (8 * k);
        }
    *reinterpret_cast<uint32_t*>(dst + (int64_t)blockIdx.z * dfstride + (int64_t)y * pitch + x4) = v;
}

// Ingest with the reader's geometry fused in (src/io/image_sequence_reader.cc): destination
// pixel (r, c) of the upright frame comes from source pixel
//   rotate   0: (r, c)          90: cv::flip(raw.t(), 0) -> (c, sw-1-r)
//          180: (sh-1-r, sw-1-c)  270: cv::flip(raw.t(), 1) -> (sh-1-c, r)           (:186-205)
// after the optional vertical / horizontal flip of the wrapper source (:53-58, :212-222) has been
// undone on (r, c).  cn = 1 copies grey, 3 / 4 applies CV_RGB2GRAY / CV_BGR2GRAY (Tracking.cc:247-260).
__global__ __launch_bounds__(256) void k_ingest(const uint8_t* __restrict__ src, int stride, int64_t sfstride,
                                                 int sw, int sh, int cn, int rIdx, int rot, int vflip, int hflip,
                                                 uint8_t* __restrict__ dst, int pitch, int64_t dfstride, int w, int h)
{
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= w || y >= h) return;
    const uint8_t* s0 = src + (int64_t)blockIdx.z * sfstride;
    const int r1 = vflip ? h - 1 - y : y;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = x4 + k;
        if (c >= w) break;
        const int c1 = hflip ? w - 1 - c : c;
        int sr, sc;
        if (rot == 0) { sr = r1; sc = c1; }
        else if (rot == 1) { sr = c1; sc = sw - 1 - r1; }
        else if (rot == 2) { sr = sh - 1 - r1; sc = sw - 1 - c1; }
        else { sr = sh - 1 - c1; sc = r1; }
        const uint8_t* px = s0 + (int64_t)sr * stride + (int64_t)sc * cn;
        int g;
        if (cn == 1) g = px[0];
        else { const int R = px[rIdx], G = px[1], B = px[2 - rIdx]; g = (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14; }
        v |= (uint32_t)g << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(dst + (int64_t)blockIdx.z * dfstride + (int64_t)y * pitch + x4) = v;
}

// The row-preserving cases (rotation 0 / 180) with dword-aligned rows and w % 4 == 0: a lane's 4 destination pixels come
// from 4 CONSECUTIVE source pixels (mirrored or not), i.e. CN aligned dwords instead of 4 * CN byte loads -- the generic
// kernel moved 1.06 GB per 128-frame 1080p RGB batch at 2.1 TB/s (tools/next_tier_bench.py).  Same arithmetic.
template <int CN>
__global__ __launch_bounds__(256) void k_ingest_rows(const uint8_t* __restrict__ src, int stride, int64_t sfstride,
                                                      int k0, int k2, int mirror, int vmirror,
                                                      uint8_t* __restrict__ dst, int pitch, int64_t dfstride, int w, int h)
{
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= w || y >= h) return;
    const int sy = vmirror ? h - 1 - y : y;
    const int sx = mirror ? w - 4 - x4 : x4;                          // first of the 4 source pixels
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (int64_t)blockIdx.z * sfstride + (int64_t)sy * stride + (int64_t)sx * CN);
    uint32_t g[4];
    if (CN == 1) {
        const uint32_t v = p[0];
        g[0] = v & 0xFF; g[1] = (v >> 8) & 0xFF; g[2] = (v >> 16) & 0xFF; g[3] = v >> 24;
    } else {
        uint32_t d[CN];
#pragma unroll
        for (int i = 0; i < CN; i++) d[i] = p[i];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // channel c of pixel k is byte CN * k + c of the lane's CN dwords; (k0, k2) are the weights of channels 0 and 2
            // (R 4899 / B 1868 in the frame's channel order), G is always channel 1 (OpenCV 2.4 RGB2Gray<uchar>, shift 14)
            const int b0 = CN * k, b1 = CN * k + 1, b2 = CN * k + 2;
            const uint32_t c0 = (d[b0 >> 2] >> (8 * (b0 & 3))) & 0xFF, c1 = (d[b1 >> 2] >> (8 * (b1 & 3))) & 0xFF,
                           c2 = (d[b2 >> 2] >> (8 * (b2 & 3))) & 0xFF;
            g[k] = (c0 * (uint32_t)k0 + c1 * 9617u + c2 * (uint32_t)k2 + 8192u) >> 14;
        }
    }
    const uint32_t v = mirror ? (g[3] | (g[2] << 8) | (g[1] << 16) | (g[0] << 24)) : (g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24));
    *reinterpret_cast<uint32_t*>(dst + (int64_t)blockIdx.z * dfstride + (int64_t)y * pitch + x4) = v;
}

void pg_launch_ingest(const PgPlan& P, const uint8_t* src, int stride, int64_t fstride, int sw, int sh, int channels,
                      int rgb_order, int rot, int vflip, int hflip, int nframes, hipStream_t s)
{
    const PgLevel& L = P.lvl[0];
    dim3 block(64, 4), grid((L.w + 255) / 256, (L.h + 3) / 4, nframes);
    static const bool generic = getenv("PGORB_INGEST_GENERIC") != nullptr;
    if (!generic && (rot == 0 || rot == 2) && L.w % 4 == 0 && stride % 4 == 0 && fstride % 4 == 0 &&
        reinterpret_cast<uintptr_t>(src) % 4 == 0 && (channels == 1 || channels == 3 || channels == 4)) {
        // rotation 180 = both mirrors; the wrapper source's flips are undone on top (see k_ingest)
        const int mirror = (hflip ? 1 : 0) ^ (rot == 2), vmirror = (vflip ? 1 : 0) ^ (rot == 2);
        const int k0 = rgb_order ? 4899 : 1868, k2 = rgb_order ? 1868 : 4899;
#define PG_INGEST_ROWS(CN) hipLaunchKernelGGL(k_ingest_rows<CN>, grid, block, 0, s, src, stride, fstride, k0, k2, mirror, vmirror, \
                                              L.img, L.pitch, L.fstride, L.w, L.h)
        if (channels == 1) PG_INGEST_ROWS(1); else if (channels == 3) PG_INGEST_ROWS(3); else PG_INGEST_ROWS(4);
#undef PG_INGEST_ROWS
        return;
    }
    hipLaunchKernelGGL(k_ingest, grid, block, 0, s, src, stride, fstride, sw, sh, channels, rgb_order ? 0 : 2, rot, vflip, hflip,
                       L.img, L.pitch, L.fstride, L.w, L.h);
}

void pg_launch_color_to_gray(const PgPlan& P, const uint8_t* src, int stride, int64_t fstride, int channels,
                             int rgb_order, int nframes, hipStream_t s)
{
    const PgLevel& L = P.lvl[0];
    dim3 block(64, 4), grid((L.w + 255) / 256, (L.h + 3) / 4, nframes);
    hipLaunchKernelGGL(k_color_to_gray, grid, block, 0, s, src, stride, fstride, channels, rgb_order ? 0 : 2,
                       L.img, L.pitch, L.fstride, L.w, L.h);
}

void pg_launch_copy_level0(const PgPlan& P, const uint8_t* src, int stride, int64_t fstride,
                           int nframes, hipStream_t s)
{
    const PgLevel& L = P.lvl[0];
    dim3 block(64, 4), grid((L.w + 255) / 256, (L.h + 3) / 4, nframes);
    hipLaunchKernelGGL(k_copy_level0, grid, block, 0, s, src, stride, fstride, L.img, L.pitch,
                       L.fstride, L.w, L.h);
}

__global__ __launch_bounds__(256) void k_pyr_resize_bilinear_u8(
    const uint8_t* __restrict__ src, int spitch, int64_t sfstride,
    uint8_t* __restrict__ dst, int dpitch, int64_t dfstride, int dw, int dh,
    const int32_t* __restrict__ xofs, const int32_t* __restrict__ xofs1,
    const int16_t* __restrict__ xalpha, const int32_t* __restrict__ yofs,
    const int16_t* __restrict__ ybeta)
{
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= dw || dy >= dh) return;
    const uint8_t* s0 = src + (int64_t)blockIdx.z * sfstride + (int64_t)yofs[2 * dy] * spitch;
    const uint8_t* s1 = src + (int64_t)blockIdx.z * sfstride + (int64_t)yofs[2 * dy + 1] * spitch;
    const int b0 = ybeta[2 * dy], b1 = ybeta[2 * dy + 1];
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = x4 + k;
        if (dx < dw) {
            const int sx = xofs[dx], sx1 = xofs1[dx];
            const int a0 = xalpha[2 * dx], a1 = xalpha[2 * dx + 1];
            const int H0 = s0[sx] * a0 + s0[sx1] * a1;          // HResizeLinear, 11-bit
            const int H1 = s1[sx] * a0 + s1[sx1] * a1;
            const int v = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 0xFF) << (8 * k);
        }
    }
    *reinterpret_cast<uint32_t*>(dst + (int64_t)blockIdx.z * dfstride + (int64_t)dy * dpitch + x4) = out;
}

// Fast path (scale factors up to ~2): one lane owns 4 adjacent destination columns for
// PYR_ROWS consecutive destination rows.  Its column taps live in registers (host-built
// PgQuadTab: first source dword, per-pixel byte offsets, 11-bit coefficients); per source row it
// issues three aligned 32-bit loads and extracts the taps with v_alignbyte; the horizontal
// result of a source row is kept in registers and reused by the next destination row
// (a 1.2x down-scale needs 1.2 source rows per destination row instead of 2).
#define PYR_ROWS 8

__device__ __forceinline__ void pyr_hrow(const uint8_t* __restrict__ row, int base_dw, int last_dw,
                                         uint32_t offs, const int a0[4], const int a1[4], int H[4])
{
    const uint32_t* r = reinterpret_cast<const uint32_t*>(row);
    const uint32_t w0 = r[min(base_dw, last_dw)], w1 = r[min(base_dw + 1, last_dw)],
                   w2 = r[min(base_dw + 2, last_dw)];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t o = (offs >> (4 * k)) & 15u;
        const uint32_t lo = o < 4 ? w0 : (o < 8 ? w1 : w2);
        const uint32_t hi = o < 4 ? w1 : (o < 8 ? w2 : 0u);
        const uint32_t t = __builtin_amdgcn_alignbyte(hi, lo, o & 3u);
        H[k] = (int)(t & 0xFF) * a0[k] + (int)((t >> 8) & 0xFF) * a1[k];      // HResizeLinear, 11-bit
    }
}

__global__ __launch_bounds__(256) void k_pyr_resize_quads(
    const uint8_t* __restrict__ src, int spitch, int64_t sfstride, int sw,
    uint8_t* __restrict__ dst, int dpitch, int64_t dfstride, int dw, int dh,
    const PgQuadTab* __restrict__ qtab, const int32_t* __restrict__ yofs,
    const int16_t* __restrict__ ybeta)
{
    const int quad = blockIdx.x * 64 + threadIdx.x;
    const int dy0 = (blockIdx.y * 4 + threadIdx.y) * PYR_ROWS;
    if (quad * 4 >= dw || dy0 >= dh) return;
    const PgQuadTab T = qtab[quad];
    int a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { a0[k] = T.a0[k]; a1[k] = T.a1[k]; }
    const int last_dw = (sw - 1) >> 2;
    const uint8_t* sbase = src + (int64_t)blockIdx.z * sfstride;
    uint8_t* dbase = dst + (int64_t)blockIdx.z * dfstride + quad * 4;
    int rowA = -1, rowB = -1, HA[4] = {0, 0, 0, 0}, HB[4] = {0, 0, 0, 0};
    const int dyEnd = min(dy0 + PYR_ROWS, dh);
    for (int dy = dy0; dy < dyEnd; dy++) {
        const int sy0 = yofs[2 * dy], sy1 = yofs[2 * dy + 1];          // wave-uniform
        if (sy0 != rowA) {
            if (sy0 == rowB) {
#pragma unroll
                for (int k = 0; k < 4; k++) HA[k] = HB[k];
            } else {
                pyr_hrow(sbase + (int64_t)sy0 * spitch, T.base_dw, last_dw, T.offs, a0, a1, HA);
            }
            rowA = sy0;
        }
        if (sy1 != rowB) {
            if (sy1 == rowA) {
#pragma unroll
                for (int k = 0; k < 4; k++) HB[k] = HA[k];
            } else {
                pyr_hrow(sbase + (int64_t)sy1 * spitch, T.base_dw, last_dw, T.offs, a0, a1, HB);
            }
            rowB = sy1;
        }
        const int b0 = ybeta[2 * dy], b1 = ybeta[2 * dy + 1];
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int v = (((b0 * (HA[k] >> 4)) >> 16) + ((b1 * (HB[k] >> 4)) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 0xFF) << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(dbase + (int64_t)dy * dpitch) = out;
    }
}

// Fast path for down-scales up to 1.25x (the ORB pyramid's 1.2): a lane owns 4 destination
// columns x 4 destination rows.
//  * Those rows need at most 6 consecutive source rows (row r0(d) is base+d or base+d+1, r1 is r0
//    or r0+1 -- verified on the host, yrel[]), so all loads are issued before anything waits.
//  * Per source row ONE unaligned 8-byte load starting at the quad's first tap covers all eight
//    taps; v_perm_b32 with a per-lane selector packs (tap0, tap1) of a pixel into two u16 and
//    v_dot2_u32_u16 against the packed 11-bit coefficients is HResizeLinear -- 2 instructions
//    per pixel and row.
//  * The vertical blend picks its two rows with wave-uniform branches (4 patterns).
#include "pyramid_rows4.inc"

// grid (tiles per frame rounded up to a multiple of 8, 1, frames), block (64, 4): a tile is 256
// columns x 32 rows, a wave (one threadIdx.y) makes two groups of 4 rows.  Workgroup b lands on
// XCD b % 8; XCD k takes the k-th eighth of the frame's tiles in (row, column) order, so
// vertically adjacent tiles -- which share source rows -- read them through the same L2.
// The wave is latency bound (kernarg -> tables -> source rows -> store): all kernel arguments
// come in one scalar batch, the two groups' 32-byte records in one s_load_dwordx16 while the
// quad table load is in flight, and the 12 source-row loads of both groups are issued together.
__global__ __launch_bounds__(256) void k_pyr_resize_rows4(
    const uint8_t* __restrict__ src, int spitch, int64_t sfstride, int sh,
    uint8_t* __restrict__ dst, int dpitch, int64_t dfstride, int dw, int dh,
    const PgQuadTab2* __restrict__ qtab, const PgRowGrp* __restrict__ rowgrp, int nx, uint32_t nxMagic)
{
    asm volatile("" :: "s"(src), "s"(spitch), "s"(sfstride), "s"(sh), "s"(dst), "s"(dpitch), "s"(dfstride),
                 "s"(dw), "s"(dh), "s"(qtab), "s"(rowgrp), "s"(nx), "s"(nxMagic));
    const int t = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int ty = (nx == 1) ? t : (int)__umulhi((uint32_t)t, nxMagic), tx = t - ty * nx;      // t / nx, t % nx
    const int quad = tx * 64 + threadIdx.x;
    const int grp = (ty * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * 2;      // groups grp, grp + 1
    const int dy0 = grp * 4;
    const int ngrp = (dh + 3) >> 2;
    const PgRowGrp* rp = rowgrp + min(grp, ngrp - 1);                 // (the table has one record of slack)
    const PgQuadTab2 T = qtab[min(quad, ((dw + 3) >> 2) - 1)];
    typedef uint32_t pg_u32x16 __attribute__((ext_vector_type(16)));
    pg_u32x16 rr;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rr) : "s"(rp) : "memory");
    if (quad * 4 >= dw || dy0 >= dh) return;
    const uint32_t ra[8] = {rr[0], rr[1], rr[2], rr[3], rr[4], rr[5], rr[6], rr[7]};
    const uint32_t rb[8] = {rr[8], rr[9], rr[10], rr[11], rr[12], rr[13], rr[14], rr[15]};
    const PgRowGrp R0 = pyr_unpack_group(ra), R1 = pyr_unpack_group(rb);
    const bool second = dy0 + 4 < dh;                                  // wave-uniform
    const uint8_t* sbase = src + (int64_t)blockIdx.z * sfstride + T.xb;
    PgU2 w0[6], w1[6];
#pragma unroll
    for (int k = 0; k < 6; k++)
        w0[k] = *reinterpret_cast<const PgU2*>(sbase + (int64_t)min(R0.sFirst + k, sh - 1) * spitch);
    const int s1 = second ? R1.sFirst : R0.sFirst;
#pragma unroll
    for (int k = 0; k < 6; k++)
        w1[k] = *reinterpret_cast<const PgU2*>(sbase + (int64_t)min(s1 + k, sh - 1) * spitch);
    uint8_t* dbase = dst + (int64_t)blockIdx.z * dfstride + quad * 4;
    pyr_group(w0, T, R0, dy0, dh, dbase, dpitch);
    if (second) pyr_group(w1, T, R1, dy0 + 4, dh, dbase, dpitch);
}

// LDS-staged variant: the workgroup's 256 x 32 destination tile first brings its source rectangle
// into LDS with LDS-DMA (16-byte chunks, each source line fetched once and in full 16-byte pieces
// instead of 64 overlapping 8-byte lane windows per row), then every lane cuts its 8-byte windows
// out of LDS (three aligned dwords + v_alignbyte).  Same arithmetic as k_pyr_resize_rows4.
typedef __attribute__((address_space(1))) const void* pg_gptr_t;
typedef __attribute__((address_space(3))) void* pg_lptr_t;

// GPW = 4-row groups per wave: the tile is 256 columns x 16 * GPW rows.  GPW = 2 is the shipped shape; 1 and 4
// exist for the tile-size sweep (PGORB_PYR_TILE_ROWS) and load their row-group records with plain
// scalar loads.
template <int GPW>
__global__ __launch_bounds__(256) void k_pyr_resize_rows4_lds(
    const uint8_t* __restrict__ src, int spitch, int64_t sfstride, int sh,
    uint8_t* __restrict__ dst, int dpitch, int64_t dfstride, int dw, int dh,
    const PgQuadTab2* __restrict__ qtab, const PgRowGrp* __restrict__ rowgrp, int nx, uint32_t nxMagic,
    const int32_t* __restrict__ tilex, int cpr, int cprInv, int rowsTile, int32_t* __restrict__ clearWord)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t pyr_lds[];      // [rowsTile][cpr * 16]
    asm volatile("" :: "s"(src), "s"(spitch), "s"(sfstride), "s"(sh), "s"(dst), "s"(dpitch), "s"(dfstride),
                 "s"(dw), "s"(dh), "s"(qtab), "s"(rowgrp), "s"(nx), "s"(nxMagic), "s"(tilex), "s"(cpr), "s"(cprInv), "s"(rowsTile), "s"(clearWord));
    // the first launch of a batch also clears the context's device status word (K2 / K3 report through it and come later in the
    // stream): a memset of its own was a 4-us launch in front of every step
    if (clearWord && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0 && threadIdx.y == 0) *clearWord = 0;
    const int t = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int ty = (nx == 1) ? t : (int)__umulhi((uint32_t)t, nxMagic), tx = t - ty * nx;
    const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int quad = tx * 64 + lane;
    const int ngrp = (dh + 3) >> 2;
    if (ty * (4 * GPW) >= ngrp) return;                                // whole workgroup: padding tile
    const int grp = ty * (4 * GPW) + wv * GPW;                         // this wave: groups grp .. grp + GPW - 1
    const int dy0 = grp * 4;
    const PgQuadTab2 T = qtab[min(quad, ((dw + 3) >> 2) - 1)];
    const PgRowGrp* rp = rowgrp + min(grp, ngrp - 1);
    typedef uint32_t pg_u32x16 __attribute__((ext_vector_type(16)));
    pg_u32x16 rr;
    int sTile, x0a;
    if (GPW == 2)
        asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dword %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(rr), "=&s"(sTile), "=&s"(x0a) : "s"(rp), "s"(rowgrp + ty * 8), "s"(tilex + tx) : "memory");
    else {
        sTile = rowgrp[ty * (4 * GPW)].sFirst;
        x0a = tilex[tx];
    }
    // stage the tile's source rectangle: cpr lanes per row, 64 / cpr rows per instruction
    {
        const int pitch = cpr * 16, rowsPer = 64 / cpr;
        const int r0 = (lane * cprInv) >> 16, ch = lane - r0 * cpr;
        const uint8_t* sb = src + (int64_t)blockIdx.z * sfstride + x0a + ch * 16;
        const bool laneOn = r0 < rowsPer && x0a + ch * 16 + 16 <= spitch;      // never past the row pitch
        for (int k = wv; k * rowsPer < rowsTile; k += 4) {
            const int r = k * rowsPer + r0;
            if (laneOn && r < rowsTile)
                __builtin_amdgcn_global_load_lds((pg_gptr_t)(sb + (int64_t)min(sTile + r, sh - 1) * spitch),
                                                 (pg_lptr_t)(pyr_lds + k * rowsPer * pitch), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    if (quad * 4 >= dw || dy0 >= dh) return;
    PgRowGrp RG[GPW];
    if (GPW == 2) {
        const uint32_t ra[8] = {rr[0], rr[1], rr[2], rr[3], rr[4], rr[5], rr[6], rr[7]};
        const uint32_t rb[8] = {rr[8], rr[9], rr[10], rr[11], rr[12], rr[13], rr[14], rr[15]};
        RG[0] = pyr_unpack_group(ra); RG[GPW - 1] = pyr_unpack_group(rb);
    } else {
#pragma unroll
        for (int gi = 0; gi < GPW; gi++) {
            const uint32_t* rw = reinterpret_cast<const uint32_t*>(rp + gi);
            const uint32_t ra[8] = {rw[0], rw[1], rw[2], rw[3], rw[4], rw[5], rw[6], rw[7]};
            RG[gi] = pyr_unpack_group(ra);
        }
    }
    const int o = T.xb - x0a;                                          // window offset in a staged row
    const uint8_t* lb = pyr_lds + (o & ~3);
    const uint32_t sh3 = (uint32_t)(o & 3);
    const int pitch = cpr * 16;
    uint8_t* dbase = dst + (int64_t)blockIdx.z * dfstride + quad * 4;
#pragma unroll
    for (int gi = 0; gi < GPW; gi++) {
        if (gi > 0 && dy0 + 4 * gi >= dh) break;
        const PgRowGrp& R = RG[gi];
        const int rel = R.sFirst - sTile;
        PgU2 w[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const uint32_t* d = reinterpret_cast<const uint32_t*>(lb + min(rel + k, rowsTile - 1) * pitch);
            const uint32_t d0 = d[0], d1 = d[1], d2 = d[2];
            w[k].x = __builtin_amdgcn_alignbyte(d1, d0, sh3);
            w[k].y = __builtin_amdgcn_alignbyte(d2, d1, sh3);
        }
        pyr_group(w, T, R, dy0 + 4 * gi, dh, dbase, dpitch);
    }
}

// clearWord: a device word this launch sets to zero (or null).  Returns false when the kernel variant taken cannot do that.
bool pg_launch_pyramid_level(const PgPlan& P, int level, int nframes, hipStream_t s, int32_t* clearWord)
{
    const PgLevel& S = P.lvl[level - 1];
    const PgLevel& D = P.lvl[level];
    static const bool noLds = getenv("PGORB_PYR_NO_LDS") != nullptr;
    if (D.qtab2 && D.yrel && D.pyrCpr > 0 && S.pitch % 16 == 0 && !noLds) {
        const int tileRows = 16 * D.pyrGpw;
        const int nx = (D.w + 255) / 256, ny = (D.h + tileRows - 1) / tileRows;
        const int tiles = (nx * ny + 7) & ~7;
        const uint32_t nxMagic = nx > 1 ? (uint32_t)(((1ull << 32) / (uint64_t)nx) + 1ull) : 0u;
        const int cprInv = 65536 / D.pyrCpr + 1;
        const size_t lds = (size_t)D.pyrRows * D.pyrCpr * 16;
        dim3 block(64, 4), grid(tiles, 1, nframes);
#define PG_PYR_LDS(G) hipLaunchKernelGGL(k_pyr_resize_rows4_lds<G>, grid, block, lds, s, S.img, S.pitch, S.fstride, S.h, D.img, D.pitch, D.fstride, \
                                         D.w, D.h, D.qtab2, D.rowgrp, nx, nxMagic, D.tilex, D.pyrCpr, cprInv, D.pyrRows, clearWord)
        if (D.pyrGpw == 1) PG_PYR_LDS(1); else if (D.pyrGpw == 4) PG_PYR_LDS(4); else PG_PYR_LDS(2);
#undef PG_PYR_LDS
        return true;
    }
    if (D.qtab2 && D.yrel) {
        const int nx = (D.w + 255) / 256, ny = (D.h + 31) / 32;
        const int tiles = (nx * ny + 7) & ~7;
        const uint32_t nxMagic = nx > 1 ? (uint32_t)(((1ull << 32) / (uint64_t)nx) + 1ull) : 0u;   // exact: tiles * nx < 2^32
        dim3 block(64, 4), grid(tiles, 1, nframes);
        hipLaunchKernelGGL(k_pyr_resize_rows4, grid, block, 0, s, S.img, S.pitch, S.fstride, S.h,
                           D.img, D.pitch, D.fstride, D.w, D.h, D.qtab2, D.rowgrp, nx, nxMagic);
        return false;
    }
    if (D.qtab) {
        dim3 block(64, 4), grid((D.w + 255) / 256, (D.h + 4 * PYR_ROWS - 1) / (4 * PYR_ROWS), nframes);
        hipLaunchKernelGGL(k_pyr_resize_quads, grid, block, 0, s, S.img, S.pitch, S.fstride, S.w,
                           D.img, D.pitch, D.fstride, D.w, D.h, D.qtab, D.yofs, D.ybeta);
        return false;
    }
    dim3 block(64, 4), grid((D.w + 255) / 256, (D.h + 3) / 4, nframes);
    hipLaunchKernelGGL(k_pyr_resize_bilinear_u8, grid, block, 0, s, S.img, S.pitch, S.fstride,
                       D.img, D.pitch, D.fstride, D.w, D.h, D.xofs, D.xofs1, D.xalpha, D.yofs,
                       D.ybeta);
    return false;
}

// pyramid.hip -- K1: bilinear pyramid chain for gfx950.
//
// Restates ORBextractor::ComputePyramid (thirdparty/orb-slam2/src/ORBextractor.cc:
// 1106-1131): level l = cv::resize(level l-1, INTER_LINEAR) (:1119), no pre-blur.  The
// 11-bit fixed-point arithmetic is OpenCV 2.4's; coefficient
// tables are built once per frame size on the host (api.hip: build_resize_tables) with the
// same float/double sequence as cv::resize.  The 19-px reflect border the reference adds
// (:1121,:1126) is never read on the monocular path, so planes are stored unpadded.
//
// HBM-bound u8 streaming kernel: each lane produces 4 adjacent destination pixels (one
// 32-bit store, 256 B per wave-row) from two source rows; the source rows are read through
// L1/L2 as bytes (a 1.2x down-scale touches 4.8 source bytes per 4 outputs per row).
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

void pg_launch_pyramid_level(const PgPlan& P, int level, int nframes, hipStream_t s)
{
    const PgLevel& S = P.lvl[level - 1];
    const PgLevel& D = P.lvl[level];
    dim3 block(64, 4), grid((D.w + 255) / 256, (D.h + 3) / 4, nframes);
    hipLaunchKernelGGL(k_pyr_resize_bilinear_u8, grid, block, 0, s, S.img, S.pitch, S.fstride,
                       D.img, D.pitch, D.fstride, D.w, D.h, D.xofs, D.xofs1, D.xalpha, D.yofs,
                       D.ybeta);
}

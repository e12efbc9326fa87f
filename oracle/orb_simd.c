/* oracle/orb_simd.c -- SIMD variants of the three primitives OpenCV 2.4.9 vectorises on x86-64.  TEST INFRASTRUCTURE
 * (bench.py's cpu_baseline leg: "kind": "port+simd"; tests/test_oracle.py proves every one of them bit-equal to the scalar
 * restatement in orb_oracle.c).
 *
 * The reference's CPU path runs ORB-SLAM2 on OpenCV 2.4.9 built with SSE2 (docker/Dockerfile:1,21).  Of the OpenCV calls on
 * this path three are vectorised there; everything else (HResizeLinear for 8U, fastAtan2, the quadtree, IC_Angle, rBRIEF) is
 * scalar in 2.4.9 as well:
 *   cv::FAST          features2d/fast.cpp FAST_t<16>: 16 pixels per vector -- ring pixels 0/4/8/12 as the pre-test, then the
 *                     25-step run-length count on bytes, corners scored one by one with cornerScore<16> (fast_score.cpp: eight
 *                     16-bit lanes); the row's tail (cols - 16 - 3 ... cols - 3) is scalar.  Called per cell at
 *                     ORBextractor.cc:809-815, i.e. ONE vector per row of a 37-px window.
 *   cv::resize        imgproc/imgwarp.cpp VResizeLinearVec_32s8u: the vertical pass (ORBextractor.cc:1119).
 *   cv::GaussianBlur  imgproc/filter.cpp RowVec_8u32s / SymmColumnVec_32s8u: both passes (ORBextractor.cc:1085).
 * The variants here use the same data-parallel decomposition with AVX2 where a wider vector applies (8 x 32-bit lanes for the
 * resize / blur passes -- an UPPER bound on what the reference's SSE2 build does, so the GPU / CPU ratio quoted from it is the
 * conservative one) and 16-byte vectors for FAST (a 37-px window row has room for exactly one).
 * Compiled with -mavx2; orc_simd_available() says whether this CPU can run them (the callers fall back to the scalar code).
 */
#include "orb_oracle.h"
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

int orc_simd_available(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

/* ------------------------------------------------------------------------ */
static const int k_ring_s[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* cornerScore<16> on eight 16-bit lanes: d[k] = v - ring[k mod 16], k = 0..24 (+7 of padding for the last load) */
static int corner_score_simd(const uint8_t* p, const int off[25], int threshold)
{
    short d[32];
    const int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = (short)(v - p[off[k]]);
    for (int k = 25; k < 32; k++) d[k] = 0;
    __m128i q0 = _mm_set1_epi16(-1000), q1 = _mm_set1_epi16(1000);
    for (int k = 0; k < 16; k += 8) {
        __m128i v0 = _mm_loadu_si128((const __m128i*)(d + k + 1)), v1 = _mm_loadu_si128((const __m128i*)(d + k + 2));
        __m128i a = _mm_min_epi16(v0, v1), b = _mm_max_epi16(v0, v1);
        for (int m = 3; m <= 8; m++) {
            v0 = _mm_loadu_si128((const __m128i*)(d + k + m));
            a = _mm_min_epi16(a, v0); b = _mm_max_epi16(b, v0);
        }
        v0 = _mm_loadu_si128((const __m128i*)(d + k));
        q0 = _mm_max_epi16(q0, _mm_min_epi16(a, v0)); q1 = _mm_min_epi16(q1, _mm_max_epi16(b, v0));
        v0 = _mm_loadu_si128((const __m128i*)(d + k + 9));
        q0 = _mm_max_epi16(q0, _mm_min_epi16(a, v0)); q1 = _mm_min_epi16(q1, _mm_max_epi16(b, v0));
    }
    /* lane j of q0 = the arc minima of the arcs that start at ring positions j and j + 8 (both polarities folded below):
     * every one of the 16 nine-arcs (and its ten-long extension by either neighbour, which cannot beat it) is covered */
    q0 = _mm_max_epi16(q0, _mm_sub_epi16(_mm_setzero_si128(), q1));
    q0 = _mm_max_epi16(q0, _mm_unpackhi_epi64(q0, q0));
    q0 = _mm_max_epi16(q0, _mm_srli_si128(q0, 4));
    q0 = _mm_max_epi16(q0, _mm_srli_si128(q0, 2));
    const int s = (short)_mm_cvtsi128_si32(q0);
    return (s > threshold ? s : threshold) - 1;
}

/* scalar decision + score of one pixel: orb_oracle.c's (fast_is_corner + fast_corner_score), for the row tails */
static int score_pixel_scalar(const uint8_t* p, const int off[25], int t)
{
    const int v = p[0];
    {   /* any 9-arc holds a pixel of every opposite pair: cheap reject (same result; orb_oracle.c does the same) */
        const int lo = v - t, hi = v + t;
        const int r0 = p[off[0]], r8 = p[off[8]], r4 = p[off[4]], r12 = p[off[12]];
        if (!(((r0 < lo || r8 < lo) && (r4 < lo || r12 < lo)) || ((r0 > hi || r8 > hi) && (r4 > hi || r12 > hi)))) return 0;
    }
    int d[25], run_d = 0, run_b = 0, corner = 0;
    for (int k = 0; k < 25; k++) d[k] = v - p[off[k]];
    for (int k = 0; k < 25 && !corner; k++) {
        run_d = (d[k] > t) ? run_d + 1 : 0;
        run_b = (d[k] < -t) ? run_b + 1 : 0;
        if (run_d >= 9 || run_b >= 9) corner = 1;
    }
    if (!corner) return 0;
    int a0 = t;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        for (int m = 3; m <= 8; m++) a = a < d[k + m] ? a : d[k + m];
        int c = a < d[k] ? a : d[k]; if (c > a0) a0 = c;
        c = a < d[k + 9] ? a : d[k + 9]; if (c > a0) a0 = c;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int m = 3; m <= 8; m++) b = b > d[k + m] ? b : d[k + m];
        int c = b > d[k] ? b : d[k]; if (c < b0) b0 = c;
        c = b > d[k + 9] ? b : d[k + 9]; if (c < b0) b0 = c;
    }
    return -b0 - 1;
}

/* == orc_fast9_score_map (orb_oracle.c): the corner score of every pixel, 0 where it is no FAST-9 corner at `threshold` */
void orc_fast9_score_map_simd(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score)
{
    memset(score, 0, (size_t)w * h);
    if (w < 7 || h < 7) return;
    int off[25];
    for (int k = 0; k < 25; k++) off[k] = k_ring_s[k & 15][1] * stride + k_ring_s[k & 15][0];
    const int tt = threshold < 0 ? 0 : (threshold > 255 ? 255 : threshold);
    const __m128i delta = _mm_set1_epi8((char)-128), t = _mm_set1_epi8((char)tt), K8 = _mm_set1_epi8(8);
    for (int y = 3; y < h - 3; y++) {
        const uint8_t* row = img + (size_t)y * stride;
        uint8_t* srow = score + (size_t)y * w;
        int x = 3;
        for (; x < w - 16 - 3; x += 16) {
            const uint8_t* ptr = row + x;
            __m128i v0 = _mm_loadu_si128((const __m128i*)ptr);
            const __m128i v1 = _mm_xor_si128(_mm_subs_epu8(v0, t), delta);      /* v - t, biased to signed */
            v0 = _mm_xor_si128(_mm_adds_epu8(v0, t), delta);                    /* v + t */
            const __m128i x0 = _mm_xor_si128(_mm_loadu_si128((const __m128i*)(ptr + off[0])), delta);
            const __m128i x1 = _mm_xor_si128(_mm_loadu_si128((const __m128i*)(ptr + off[4])), delta);
            const __m128i x2 = _mm_xor_si128(_mm_loadu_si128((const __m128i*)(ptr + off[8])), delta);
            const __m128i x3 = _mm_xor_si128(_mm_loadu_si128((const __m128i*)(ptr + off[12])), delta);
            /* a 9-arc holds two ADJACENT compass points: brighter (> v + t) or darker (< v - t) in one of the four adjacent pairs */
            __m128i m0 = _mm_and_si128(_mm_cmpgt_epi8(x0, v0), _mm_cmpgt_epi8(x1, v0));
            __m128i m1 = _mm_and_si128(_mm_cmpgt_epi8(v1, x0), _mm_cmpgt_epi8(v1, x1));
            m0 = _mm_or_si128(m0, _mm_and_si128(_mm_cmpgt_epi8(x1, v0), _mm_cmpgt_epi8(x2, v0)));
            m1 = _mm_or_si128(m1, _mm_and_si128(_mm_cmpgt_epi8(v1, x1), _mm_cmpgt_epi8(v1, x2)));
            m0 = _mm_or_si128(m0, _mm_and_si128(_mm_cmpgt_epi8(x2, v0), _mm_cmpgt_epi8(x3, v0)));
            m1 = _mm_or_si128(m1, _mm_and_si128(_mm_cmpgt_epi8(v1, x2), _mm_cmpgt_epi8(v1, x3)));
            m0 = _mm_or_si128(m0, _mm_and_si128(_mm_cmpgt_epi8(x3, v0), _mm_cmpgt_epi8(x0, v0)));
            m1 = _mm_or_si128(m1, _mm_and_si128(_mm_cmpgt_epi8(v1, x3), _mm_cmpgt_epi8(v1, x0)));
            if (_mm_movemask_epi8(_mm_or_si128(m0, m1)) == 0) continue;
            /* run lengths of "brighter" / "darker" along the 25-step ring walk, sixteen pixels at once */
            __m128i c0 = _mm_setzero_si128(), c1 = c0, max0 = c0, max1 = c0;
            for (int k = 0; k < 25; k++) {
                const __m128i xx = _mm_xor_si128(_mm_loadu_si128((const __m128i*)(ptr + off[k])), delta);
                m0 = _mm_cmpgt_epi8(xx, v0); m1 = _mm_cmpgt_epi8(v1, xx);
                c0 = _mm_and_si128(_mm_sub_epi8(c0, m0), m0); c1 = _mm_and_si128(_mm_sub_epi8(c1, m1), m1);
                max0 = _mm_max_epu8(max0, c0); max1 = _mm_max_epu8(max1, c1);
            }
            max0 = _mm_max_epu8(max0, max1);
            int m = _mm_movemask_epi8(_mm_cmpgt_epi8(max0, K8));
            for (int k = 0; m > 0 && k < 16; k++, m >>= 1)
                if (m & 1) srow[x + k] = (uint8_t)corner_score_simd(ptr + k, off, threshold);
        }
        for (; x < w - 3; x++) srow[x] = (uint8_t)score_pixel_scalar(row + x, off, threshold);
    }
}

/* == orc_fast9_nms (orb_oracle.c) on the SIMD score map */
int orc_fast9_nms_simd(const uint8_t* img, int w, int h, int stride, int threshold, orc_cand* out, int cap)
{
    if (w < 7 || h < 7) return 0;
    uint8_t* score = (uint8_t*)malloc((size_t)w * h);
    orc_fast9_score_map_simd(img, w, h, stride, threshold, score);
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const int s = score[(size_t)y * w + x];
            if (!s) continue;
            const uint8_t* r0 = score + (size_t)(y - 1) * w + x;
            const uint8_t* r1 = score + (size_t)y * w + x;
            const uint8_t* r2 = score + (size_t)(y + 1) * w + x;
            if (s > r1[1] && s > r1[-1] && s > r0[-1] && s > r0[0] && s > r0[1] && s > r2[-1] && s > r2[0] && s > r2[1]) {
                if (n < cap) { out[n].x = x; out[n].y = y; out[n].response = s; }
                n++;
            }
        }
    free(score);
    return n;
}

/* ------------------------------------------------------------------------ */
/* VResizeLinear for 8U, eight pixels per step: ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2 */
void orc_vresize_row_simd(const int* r0, const int* r1, int b0, int b1, uint8_t* out, int n)
{
    const __m256i vb0 = _mm256_set1_epi32(b0), vb1 = _mm256_set1_epi32(b1), two = _mm256_set1_epi32(2);
    int x = 0;
    for (; x + 8 <= n; x += 8) {
        const __m256i a = _mm256_srai_epi32(_mm256_loadu_si256((const __m256i*)(r0 + x)), 4);
        const __m256i b = _mm256_srai_epi32(_mm256_loadu_si256((const __m256i*)(r1 + x)), 4);
        __m256i v = _mm256_add_epi32(_mm256_srai_epi32(_mm256_mullo_epi32(a, vb0), 16), _mm256_srai_epi32(_mm256_mullo_epi32(b, vb1), 16));
        v = _mm256_srai_epi32(_mm256_add_epi32(v, two), 2);
        const __m256i p16 = _mm256_packus_epi32(v, v);                          /* per 128-bit half: 4 values twice */
        const __m256i p8 = _mm256_packus_epi16(p16, p16);
        const int lo = _mm256_extract_epi32(p8, 0), hi = _mm256_extract_epi32(p8, 4);
        memcpy(out + x, &lo, 4); memcpy(out + x + 4, &hi, 4);
    }
    for (; x < n; x++) out[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
}

/* ------------------------------------------------------------------------ */
/* 7-tap symmetric row pass of the blur on interior pixels [3, w - 3): acc = K0 (s[-3] + s[3]) + K1 (s[-2] + s[2]) + K2 (s[-1] + s[1]) + K3 s[0] */
void orc_blur_row_simd(const uint8_t* s, int w, const int K[7], int* R)
{
    const __m256i k0 = _mm256_set1_epi32(K[0]), k1 = _mm256_set1_epi32(K[1]), k2 = _mm256_set1_epi32(K[2]), k3 = _mm256_set1_epi32(K[3]);
    int x = 3;
#define LD8(p) _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i*)(p)))
    for (; x + 8 <= w - 3; x += 8) {
        __m256i acc = _mm256_mullo_epi32(k0, _mm256_add_epi32(LD8(s + x - 3), LD8(s + x + 3)));
        acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(k1, _mm256_add_epi32(LD8(s + x - 2), LD8(s + x + 2))));
        acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(k2, _mm256_add_epi32(LD8(s + x - 1), LD8(s + x + 1))));
        acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(k3, LD8(s + x)));
        _mm256_storeu_si256((__m256i*)(R + x), acc);
    }
#undef LD8
    for (; x + 3 < w; x++)
        R[x] = K[0] * (s[x - 3] + s[x + 3]) + K[1] * (s[x - 2] + s[x + 2]) + K[2] * (s[x - 1] + s[x + 1]) + K[3] * s[x];
}

/* column pass of the blur for x in [0, n): C = K0 (r0 + r6) + K1 (r1 + r5) + K2 (r2 + r4) + K3 r3, v = (C + 32768) >> 16, a tie
 * (low half 0x8000) to EVEN where x < wvec and tie_mode == 0 (what cvtps2dq does in SymmColumnVec_32s8u), saturated to 255 */
void orc_blur_col_simd(const int* const rr[7], int n, const int K[7], int tie_mode, int wvec, uint8_t* d)
{
    const __m256i k0 = _mm256_set1_epi32(K[0]), k1 = _mm256_set1_epi32(K[1]), k2 = _mm256_set1_epi32(K[2]), k3 = _mm256_set1_epi32(K[3]);
    const __m256i half = _mm256_set1_epi32(32768), lo16 = _mm256_set1_epi32(0xFFFF), one = _mm256_set1_epi32(1), c255 = _mm256_set1_epi32(255);
    int x = 0;
#define LD(i) _mm256_loadu_si256((const __m256i*)(rr[i] + x))
    for (; x + 8 <= n; x += 8) {
        __m256i C = _mm256_mullo_epi32(k0, _mm256_add_epi32(LD(0), LD(6)));
        C = _mm256_add_epi32(C, _mm256_mullo_epi32(k1, _mm256_add_epi32(LD(1), LD(5))));
        C = _mm256_add_epi32(C, _mm256_mullo_epi32(k2, _mm256_add_epi32(LD(2), LD(4))));
        C = _mm256_add_epi32(C, _mm256_mullo_epi32(k3, LD(3)));
        __m256i v = _mm256_srai_epi32(_mm256_add_epi32(C, half), 16);
        if (tie_mode == 0) {
            /* lanes with x < wvec (wvec is a multiple of 4: a vector of 8 may straddle it) */
            const __m256i idx = _mm256_add_epi32(_mm256_set1_epi32(x), _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7));
            const __m256i inv = _mm256_cmpgt_epi32(_mm256_set1_epi32(wvec), idx);
            const __m256i tie = _mm256_and_si256(_mm256_cmpeq_epi32(_mm256_and_si256(C, lo16), half), inv);
            v = _mm256_andnot_si256(_mm256_and_si256(tie, one), v);
        }
        v = _mm256_min_epi32(v, c255);                                          /* (C >= 0: no lower clamp needed) */
        const __m256i p16 = _mm256_packus_epi32(v, v);
        const __m256i p8 = _mm256_packus_epi16(p16, p16);
        const int lo = _mm256_extract_epi32(p8, 0), hi = _mm256_extract_epi32(p8, 4);
        memcpy(d + x, &lo, 4); memcpy(d + x + 4, &hi, 4);
    }
#undef LD
    for (; x < n; x++) {
        const int C = K[0] * (rr[0][x] + rr[6][x]) + K[1] * (rr[1][x] + rr[5][x]) + K[2] * (rr[2][x] + rr[4][x]) + K[3] * rr[3][x];
        int v = (C + 32768) >> 16;
        if (tie_mode == 0 && x < wvec && (C & 0xFFFF) == 0x8000) v &= ~1;
        d[x] = (uint8_t)(v > 255 ? 255 : v);
    }
}

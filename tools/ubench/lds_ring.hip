// Micro-benchmark (round 3): the 16 ring pixels + centre of a FAST candidate out of an LDS tile (pitch 48), either as
// 17 ds_read_u8 (what K2's exact score did) or as 7 byte-UNALIGNED wide reads (rows y+-3: one b32 at x-1; rows y+-2,
// y+-1, y: one b64 at x-2 / x-3) with the bytes picked at fixed positions.  Checks that both give the same sums
// (i.e. that gfx950 serves unaligned ds_read_b32 / b64 correctly) and times them: 8 one-wave workgroups' worth of LDS
// per CU slot like K2 (4.9 KB per wave), 64 random candidates of a 31 x 31 interior per round.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 256
#define TP 48
template <int N> struct __attribute__((packed, aligned(1))) Raw { uint8_t b[N]; };
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    for (int i = threadIdx.x; i < 37 * TP / 4 + 16; i += 64) ((uint32_t*)sm)[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t h = (threadIdx.x * 2654435761u + seed + blockIdx.x) >> 4;
    uint32_t acc = 0;
    for (int it = 0; it < ITER; it++) {
        const int ix = h % 31, iy = (h >> 8) % 31;
        const uint8_t* c = sm + (iy + 3) * TP + 4 + ix;
        if (MODE == 0) {
            const int p = TP;
            acc += c[0] + 2 * c[3 * p] + 3 * c[3 * p + 1] + 5 * c[2 * p + 2] + 7 * c[p + 3] + 11 * c[3] + 13 * c[-p + 3] + 17 * c[-2 * p + 2] +
                   19 * c[-3 * p + 1] + 23 * c[-3 * p] + 29 * c[-3 * p - 1] + 31 * c[-2 * p - 2] + 37 * c[-p - 3] + 41 * c[-3] + 43 * c[p - 3] +
                   47 * c[2 * p - 2] + 53 * c[3 * p - 1];
        } else {
            const uint32_t u3 = ld32(c + 3 * TP - 1), d3 = ld32(c - 3 * TP - 1);             // x-1, x, x+1
            const uint64_t u2 = ld64(c + 2 * TP - 2), d2 = ld64(c - 2 * TP - 2);             // x-2 .. x+2 in bytes 0, 4
            const uint64_t u1 = ld64(c + TP - 3), d1 = ld64(c - TP - 3), m = ld64(c - 3);    // x-3 .. x+3 in bytes 0, 6 (centre byte 3)
#define B(v, k) ((uint32_t)((v) >> (8 * (k))) & 0xFF)
            acc += B(m, 3) + 2 * B(u3, 1) + 3 * B(u3, 2) + 5 * B(u2, 4) + 7 * B(u1, 6) + 11 * B(m, 6) + 13 * B(d1, 6) + 17 * B(d2, 4) +
                   19 * B(d3, 2) + 23 * B(d3, 1) + 29 * B(d3, 0) + 31 * B(d2, 0) + 37 * B(d1, 0) + 41 * B(m, 0) + 43 * B(u1, 0) +
                   47 * B(u2, 0) + 53 * B(u3, 0);
        }
        h = h * 1664525u + 1013904223u;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int MODE> double run(const char* name, uint32_t* d, int ninst)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 32 * 8;
    k<MODE><<<blocks, 64, 4912>>>(d, 1);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 64, 4912>>>(d, 2);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double rounds_per_cu = 32.0 * 8 * ITER;
    printf("%-40s %7.3f ms -> %.1f cycles per 64-candidate round per CU (@2.4 GHz), %d DS instructions per round\n", name, ms,
           ms * 1e-3 * 2.4e9 / rounds_per_cu, ninst);
    return ms;
}
int main()
{
    const int n = 256 * 32 * 8 * 64;
    uint32_t *d0, *d1; (void)hipMalloc(&d0, n * 4); (void)hipMalloc(&d1, n * 4);
    run<0>("17 x ds_read_u8", d0, 17);
    run<1>("2 x b32 + 5 x b64, byte-unaligned", d1, 7);
    uint32_t* h0 = (uint32_t*)malloc(n * 4); uint32_t* h1 = (uint32_t*)malloc(n * 4);
    (void)hipMemcpy(h0, d0, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h1, d1, n * 4, hipMemcpyDeviceToHost);
    long bad = 0; for (int i = 0; i < n; i++) bad += h0[i] != h1[i];
    printf("unaligned wide reads %s the byte reads (%ld of %d sums differ)\n", bad ? "DIFFER FROM" : "equal", bad, n);
    return bad != 0;
}

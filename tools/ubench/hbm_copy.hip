// Micro-benchmark (round 5): what a hand-written 16-byte-per-lane copy reaches on this box (read + write), plain and with
// non-temporal loads / stores, at several grid sizes; plus read-only and write-only streams.  Reference point for K1 (4.8 TB/s)
// next to torch's copy kernel (tools/experiments/hbm_copy_rate.py: 5.0 TB/s) and the guide's 6.29 TB/s float4 copy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT, int UNR>
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * UNR;
    for (size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x; i < n; i += stride) {
        f4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) if (i + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(a + i + u * 256) : a[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNR; u++) if (i + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * 256); else b[i + u * 256] = v[u]; }
    }
}
template <int UNR>
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ a, f4* __restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * UNR;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < UNR; u++) if (i + u * 256 < n) acc += a[i + u * 256];
    }
    if (acc.x == 12345.f) b[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(f4* __restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const f4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) b[i] = v;
}
template <typename F> double timeit(F f, int reps)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; r++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    f4 *a, *b; (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
    for (int wgPerCu : {2, 4, 8, 16, 32}) {
        const int grid = 256 * wgPerCu;
        double ms;
        ms = timeit([&] { k_copy<0, 1><<<grid, 256>>>(a, b, n); }, 10);
        printf("copy  plain  unroll 1  %4d wg/cu: %.3f ms  %.2f TB/s (read + write)\n", wgPerCu, ms, 2.0 * bytes / ms / 1e9);
        ms = timeit([&] { k_copy<0, 4><<<grid, 256>>>(a, b, n); }, 10);
        printf("copy  plain  unroll 4  %4d wg/cu: %.3f ms  %.2f TB/s\n", wgPerCu, ms, 2.0 * bytes / ms / 1e9);
        ms = timeit([&] { k_copy<1, 4><<<grid, 256>>>(a, b, n); }, 10);
        printf("copy  nt     unroll 4  %4d wg/cu: %.3f ms  %.2f TB/s\n", wgPerCu, ms, 2.0 * bytes / ms / 1e9);
        ms = timeit([&] { k_read<4><<<grid, 256>>>(a, b, n); }, 10);
        printf("read         unroll 4  %4d wg/cu: %.3f ms  %.2f TB/s\n", wgPerCu, ms, 1.0 * bytes / ms / 1e9);
        ms = timeit([&] { k_write<<<grid, 256>>>(b, n); }, 10);
        printf("write                  %4d wg/cu: %.3f ms  %.2f TB/s\n", wgPerCu, ms, 1.0 * bytes / ms / 1e9);
    }
    // one workgroup per 4 KiB (no grid-stride loop), the shape of a tile kernel
    {
        const int grid = (int)(n / 256);
        double ms = timeit([&] { k_copy<0, 1><<<grid, 256>>>(a, b, n); }, 10);
        printf("copy  plain  one 4-KiB piece per workgroup (%d workgroups): %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * bytes / ms / 1e9);
    }
    return 0;
}

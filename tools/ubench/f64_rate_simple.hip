// f64 issue cost on gfx950 (the sin / cos contract of K4-6 runs 42 f64 operations per keypoint, redundantly in all lanes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 2048
template <int MODE> __global__ void k(double* out, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < ITER; it++) {
        if (MODE == 0) { a0 = __dadd_rn(a0, c); a1 = __dadd_rn(a1, c); a2 = __dadd_rn(a2, c); a3 = __dadd_rn(a3, c); a4 = __dadd_rn(a4, c); a5 = __dadd_rn(a5, c); a6 = __dadd_rn(a6, c); a7 = __dadd_rn(a7, c); }
        if (MODE == 1) { a0 = __dmul_rn(a0, b); a1 = __dmul_rn(a1, b); a2 = __dmul_rn(a2, b); a3 = __dmul_rn(a3, b); a4 = __dmul_rn(a4, b); a5 = __dmul_rn(a5, b); a6 = __dmul_rn(a6, b); a7 = __dmul_rn(a7, b); }
        if (MODE == 2) { a0 = __fma_rn(a0, b, c); a1 = __fma_rn(a1, b, c); a2 = __fma_rn(a2, b, c); a3 = __fma_rn(a3, b, c); a4 = __fma_rn(a4, b, c); a5 = __fma_rn(a5, b, c); a6 = __fma_rn(a6, b, c); a7 = __fma_rn(a7, b, c); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE> void run(const char* name)
{
    double* d; (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256 * 8, 256>>>(d, 1.0);
    (void)hipEventRecord(e0);
    k<MODE><<<256 * 8, 256>>>(d, 2.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s %7.3f ms -> %.2f cycles per wave-instruction per SIMD (@2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (256.0 * 8 * 4 * ITER * 8 / 1024.0));
}
int main() { run<0>("add_f64"); run<1>("mul_f64"); run<2>("fma_f64"); return 0; }

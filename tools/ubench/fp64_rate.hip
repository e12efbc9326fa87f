// fp64_rate.hip -- issue rate and dependent-issue latency of fp64 VALU ops on gfx950, one wave.
// Prints shader-clock cycles per operation for: a dependent v_add_f64 chain, 4 independent add chains,
// a dependent v_mul_f64 chain, a dependent v_fma_f64 chain, 4 independent fma chains.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define N 4096

template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double a0, double b0)
{
    double a = a0 + threadIdx.x, b = a0 * 2, c = a0 * 3, d = a0 * 4;
    const double x = b0;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; i++) {
        if (MODE == 0) a = a + x;
        if (MODE == 1) { a = a + x; b = b + x; c = c + x; d = d + x; }
        if (MODE == 2) a = a * x;
        if (MODE == 3) a = __builtin_fma(a, x, x);
        if (MODE == 4) { a = __builtin_fma(a, x, x); b = __builtin_fma(b, x, x); c = __builtin_fma(c, x, x); d = __builtin_fma(d, x, x); }
        if (MODE == 5) { a = a * x; b = b * x; c = c * x; d = d * x; }
    }
    asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d));
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int ops)
{
    double* o; unsigned long long* c; unsigned long long h = 0;
    hipMalloc(&o, 64 * 8); hipMalloc(&c, 8);
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, o, c, 1.0, 1.0000001); hipDeviceSynchronize(); }
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-34s %.2f cycles per op\n", name, (double)h / ((double)N * ops));
    hipFree(o); hipFree(c);
}

int main()
{
    run<0>("dependent v_add_f64 chain", 1);
    run<1>("4 independent v_add_f64 chains", 4);
    run<2>("dependent v_mul_f64 chain", 1);
    run<5>("4 independent v_mul_f64 chains", 4);
    run<3>("dependent v_fma_f64 chain", 1);
    run<4>("4 independent v_fma_f64 chains", 4);
    return 0;
}

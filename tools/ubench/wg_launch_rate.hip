// How fast does the MI355X dispatch one-wave workgroups?  k_fast_cells and k_describe launch
// 8e5 / 2.6e5 of them per step.  Build: hipcc --offload-arch=gfx950 -O2 -o wg_launch_rate wg_launch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_lds(int* p) { extern __shared__ int sm[]; if (p && threadIdx.x == 9999) p[0] = sm[threadIdx.x]; }
template <int N> __global__ void k_spin(int* p)       // ~N dependent VALU ops per wave
{
    int v = threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < N; i++) v = v * 3 + 1;
    if (v == 0x7fffffff && p) p[0] = v;
}
template <typename F> float timeit(F f, int reps = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main()
{
    const int n = 6344 * 128;
    for (int threads : {64, 256}) {
        const int g = n * 64 / threads;
        float t0 = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(threads), 0, 0, nullptr); });
        float t1 = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(g), dim3(threads), 4912 * threads / 64, 0, nullptr); });
        float t2 = timeit([&] { hipLaunchKernelGGL(k_spin<200>, dim3(g), dim3(threads), 0, 0, nullptr); });
        float t3 = timeit([&] { hipLaunchKernelGGL(k_spin<800>, dim3(g), dim3(threads), 0, 0, nullptr); });
        printf("%3d threads/WG, %7d WGs (%d waves): empty %.3f ms (%.0f WG/us)  +4.8KB LDS/wave %.3f ms  200 VALU %.3f ms  800 VALU %.3f ms\n",
               threads, g, n, t0, g / t0 / 1e3, t1, t2, t3);
    }
    return 0;
}

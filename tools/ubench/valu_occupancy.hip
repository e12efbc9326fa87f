// Micro-benchmark (round 6): aggregate VALU issue rate of a SIMD against the number of resident waves.  valu_rate3/4 measured
// "2.7 cycles per fast-class wave instruction, 4.6 per slow-class" -- at what occupancy?  W one-wave workgroups per SIMD
// (W = 1, 2, 4, 8), each a loop of 8 independent chains: cycles per wave-instruction per SIMD at an assumed 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
#define BODY(ASM) \
    for (int it = 0; it < ITER; it++) { \
        asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c)); \
        asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c)); \
        asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c)); \
        asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c)); }
#define KERNEL(NAME, ASM) \
__global__ __launch_bounds__(64) void NAME(uint32_t* out, uint32_t seed) { \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    uint32_t b = seed ^ 0x5bd1e995, c = threadIdx.x | 1; \
    BODY(ASM) \
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
KERNEL(k_add, "v_add_u32_e32 %0, %0, %1")
KERNEL(k_and, "v_and_b32_e32 %0, %0, %1")
KERNEL(k_max3, "v_max3_i32 %0, %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_pkmax, "v_pk_max_u16 %0, %0, %1")
KERNEL(k_lerp, "v_lerp_u8 %0, %0, %1, %2")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
KERNEL(k_pkmax3f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
template <typename K> void run(const char* name, K kern, uint32_t* d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-14s", name);
    for (int W = 1; W <= 8; W *= 2) {
        const int blocks = 256 * 4 * W;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 1u);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 2u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)W * ITER * 8;
        printf("  W=%d: %6.3f ms %5.2f cyc/inst/SIMD", W, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
    }
    printf("\n");
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 4 << 20);
    run("v_add_u32", k_add, d); run("v_and_b32", k_and, d); run("v_max3_i32", k_max3, d); run("v_perm_b32", k_perm, d);
    run("v_pk_max_u16", k_pkmax, d); run("v_lerp_u8", k_lerp, d); run("v_bitop3_b32", k_bitop3, d); run("v_pk_max3_f16", k_pkmax3f16, d);
    return 0;
}

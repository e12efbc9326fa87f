// Micro-benchmark 4: issue rate of the integer multiplies on gfx950 (same harness as valu_rate3): the full 32-bit forms
// (v_mul_lo_u32, v_mul_hi_u32, v_mad_u64_u32) against the 24-bit ones (v_mul_u32_u24, v_mad_u32_u24, v_mul_hi_u32_u24).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048
#define BODY(ASM) \
    for (int it = 0; it < ITER; it++) { \
        asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c)); \
        asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c)); \
        asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c)); \
        asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c)); }
#define KERNEL(NAME, ASM) \
__global__ void NAME(uint32_t* out, uint32_t seed) { \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    uint32_t b = seed ^ 0x5bd1e995, c = threadIdx.x | 1; \
    BODY(ASM) \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
KERNEL(k_add, "v_add_u32_e32 %0, %0, %1")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_mul_u24, "v_mul_u32_u24_e32 %0, %0, %1")
KERNEL(k_mul_hi_u24, "v_mul_hi_u32_u24_e32 %0, %0, %1")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(k_dot2, "v_dot2_u32_u16 %0, %0, %1, %2")
__global__ void k_mad_u64(uint32_t* out, uint32_t seed)
{
    typedef unsigned long long u64;
    u64 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed ^ 0x5bd1e995, c = threadIdx.x | 1;
#define M64(A) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(b), "v"(c) : "vcc")
    for (int it = 0; it < ITER; it++) { M64(a0); M64(a1); M64(a2); M64(a3); M64(a4); M64(a5); M64(a6); M64(a7); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
template <typename F> void run(const char* name, F f)
{
    uint32_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f<<<256 * 8, 256>>>(d, 1);
    (void)hipEventRecord(e0);
    f<<<256 * 8, 256>>>(d, 2);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = 256.0 * 8 * 4 * ITER * 8 / 1024.0;
    printf("%-18s %7.3f ms -> %.2f cycles per wave-instruction per SIMD (@2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    (void)hipFree(d);
}
int main()
{
    run("add_u32 e32", k_add); run("mul_lo_u32", k_mul_lo); run("mul_hi_u32", k_mul_hi); run("mul_u32_u24", k_mul_u24);
    run("mul_hi_u32_u24", k_mul_hi_u24); run("mad_u32_u24", k_mad_u24); run("mad_i32_i24", k_mad_i24);
    run("lshl_add_u32", k_lshl_add); run("dot2_u32_u16", k_dot2); run("mad_u64_u32", k_mad_u64);
    return 0;
}

// Micro-benchmark 2: issue cost of individual gfx950 VALU encodings (inline asm, 8 independent chains).
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
KERNEL(k_min_vop2, "v_min_u32_e32 %0, %0, %1")
KERNEL(k_min_sdwa, "v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2")
KERNEL(k_add_vop2, "v_add_u32_e32 %0, %0, %1")
KERNEL(k_min3, "v_min3_u32 %0, %0, %1, %2")
KERNEL(k_pk_min, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_pk_add, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL(k_xor, "v_xor_b32_e32 %0, %0, %1")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL(k_mul24, "v_mul_u32_u24_e32 %0, %0, %1")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_dot2, "v_dot2_u32_u16 %0, %0, %1, %2")
KERNEL(k_sad, "v_sad_u8 %0, %0, %1, %2")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
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
    printf("%-14s %7.3f ms -> %.2f cycles per wave-instruction per SIMD (@2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    (void)hipFree(d);
}
int main()
{
    run("min_u32 VOP2", k_min_vop2); run("min_u32 SDWA", k_min_sdwa); run("add_u32 VOP2", k_add_vop2);
    run("min3_u32", k_min3); run("pk_min_u16", k_pk_min); run("pk_add_u16", k_pk_add); run("perm_b32", k_perm);
    run("bfe_u32", k_bfe); run("and_or_b32", k_and_or); run("lshl_or_b32", k_lshl_or); run("xor VOP2", k_xor);
    run("bcnt_u32_b32", k_bcnt); run("mul_u32_u24", k_mul24); run("mad_u32_u24", k_mad24); run("mul_lo_u32", k_mullo);
    run("dot2_u32_u16", k_dot2); run("sad_u8", k_sad); run("alignbyte", k_alignbyte);
    return 0;
}

// Micro-benchmark 3: which gfx950 VALU encodings issue at the fast rate?  valu_rate2 showed v_add_u32 / v_xor_b32 (VOP2)
// at ~2.7 cycles per wave-instruction per SIMD and everything VOP3 / packed at ~4.6.  8 independent chains, inline asm.
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
KERNEL(k_sub, "v_sub_u32_e32 %0, %0, %1")
KERNEL(k_and, "v_and_b32_e32 %0, %0, %1")
KERNEL(k_or, "v_or_b32_e32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32_e32 %0, 1, %0")
KERNEL(k_lshr, "v_lshrrev_b32_e32 %0, 1, %0")
KERNEL(k_maxu, "v_max_u32_e32 %0, %0, %1")
KERNEL(k_mini, "v_min_i32_e32 %0, %0, %1")
KERNEL(k_minu16, "v_min_u16_e32 %0, %0, %1")
KERNEL(k_addu16, "v_add_u16_e32 %0, %0, %1")
KERNEL(k_subu16, "v_sub_u16_e32 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32_e32 %0, %1")
KERNEL(k_add_e64, "v_add_u32_e64 %0, %0, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_pk_add, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_sub, "v_pk_sub_u16 %0, %0, %1")
KERNEL(k_pk_max, "v_pk_max_u16 %0, %0, %1")
KERNEL(k_pk_lshl, "v_pk_lshlrev_b16 %0, 1, %0")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
KERNEL(k_max3, "v_max3_i32 %0, %0, %1, %2")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
KERNEL(k_sad_u16, "v_sad_u16 %0, %0, %1, %2")
KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(k_addc, "v_addc_co_u32_e32 %0, vcc, %0, %1, vcc")
KERNEL(k_cndmask, "v_cndmask_b32_e32 %0, %0, %1, vcc")
KERNEL(k_dpp_add, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_sdwa_add, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
KERNEL(k_sdwa_sub, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2")
KERNEL(k_add_f32, "v_add_f32_e32 %0, %0, %1")
KERNEL(k_max_f32, "v_max_f32_e32 %0, %0, %1")
KERNEL(k_min_f16, "v_min_f16_e32 %0, %0, %1")
KERNEL(k_pk_min_f16, "v_pk_min_f16 %0, %0, %1")
KERNEL(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_min3_f32, "v_min3_f32 %0, %0, %1, %2")
KERNEL(k_mul_f32, "v_mul_f32_e32 %0, %0, %1")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_rndne_f32, "v_rndne_f32_e32 %0, %0")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32_e32 %0, %0")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32_e32 %0, %0")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 2")
KERNEL(k_bfe_i, "v_bfe_i32 %0, %0, 4, 8")
KERNEL(k_cmp, "v_cmp_gt_u32_e32 vcc, %0, %1")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
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
    run("add_u32 e32", k_add); run("sub_u32 e32", k_sub); run("and_b32 e32", k_and); run("or_b32 e32", k_or);
    run("lshlrev e32", k_lshl); run("lshrrev e32", k_lshr); run("max_u32 e32", k_maxu); run("min_i32 e32", k_mini);
    run("min_u16 e32", k_minu16); run("add_u16 e32", k_addu16); run("sub_u16 e32", k_subu16); run("mov_b32", k_mov);
    run("add_u32 e64", k_add_e64); run("add3_u32", k_add3); run("or3_b32", k_or3);
    run("pk_add_u16", k_pk_add); run("pk_sub_u16", k_pk_sub); run("pk_max_u16", k_pk_max); run("pk_lshlrev_b16", k_pk_lshl);
    run("alignbit", k_alignbit); run("bfi", k_bfi); run("max3_i32", k_max3); run("bitop3", k_bitop3); run("sad_u16", k_sad_u16);
    run("mad_i32_i24", k_mad_i24); run("addc_co", k_addc); run("cndmask e32", k_cndmask); run("add_u32 dpp", k_dpp_add);
    run("add_u32 sdwa", k_sdwa_add); run("sub_u32 sdwa", k_sdwa_sub);
    run("add_f32 e32", k_add_f32); run("max_f32 e32", k_max_f32); run("min_f16 e32", k_min_f16); run("pk_min_f16", k_pk_min_f16);
    run("med3_f32", k_med3_f32); run("min3_f32", k_min3_f32);
    run("mul_f32 e32", k_mul_f32); run("fma_f32", k_fma_f32); run("rndne_f32", k_rndne_f32); run("cvt_i32_f32", k_cvt_i32_f32);
    run("cvt_f32_i32", k_cvt_f32_i32); run("lshl_add_u32", k_lshl_add); run("add_lshl_u32", k_add_lshl); run("bfe_i32", k_bfe_i);
    run("cmp_gt_u32 (vcc)", k_cmp); run("dot4_u32_u8", k_dot4); run("mul_hi_u32", k_mul_hi);
    return 0;
}

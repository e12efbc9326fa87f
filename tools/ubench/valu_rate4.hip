// Micro-benchmark 4 (round 3): issue cost of the byte-parallel candidates for K2's necessary test (v_lerp_u8, v_alignbyte_b32,
// v_perm_b32, v_sad_u8 ...), of literal-constant VOP2 forms and of the bit-scan / count ops of the compaction.  Same harness as valu_rate3.
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
KERNEL(k_add_lit, "v_add_u32_e32 %0, 0x76767676, %0")
KERNEL(k_and_lit, "v_and_b32_e32 %0, 0x7f7f7f7f, %0")
KERNEL(k_add_e64, "v_add_u32_e64 %0, %0, %1")
KERNEL(k_lerp, "v_lerp_u8 %0, %0, %1, %2")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
KERNEL(k_msad_u8, "v_msad_u8 %0, %0, %1, %2")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_bfe_u, "v_bfe_u32 %0, %0, 4, 8")
KERNEL(k_ffbl, "v_ffbl_b32_e32 %0, %0")
KERNEL(k_ffbh, "v_ffbh_u32_e32 %0, %0")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_not, "v_not_b32_e32 %0, %0")
KERNEL(k_xor, "v_xor_b32_e32 %0, %0, %1")
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
KERNEL(k_min3_u16, "v_min3_u16 %0, %0, %1, %2")
KERNEL(k_max_u16, "v_max_u16_e32 %0, %0, %1")
KERNEL(k_lshrrev_lit, "v_lshrrev_b32_e32 %0, 1, %0")
KERNEL(k_mul_u24, "v_mul_u32_u24_e32 %0, %0, %1")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %0, %1, %2")
KERNEL(k_sdwa_sub_b, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2")
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
    run("add", k_add);
    run("add_lit", k_add_lit);
    run("and_lit", k_and_lit);
    run("add_e64", k_add_e64);
    run("lerp", k_lerp);
    run("alignbyte", k_alignbyte);
    run("perm", k_perm);
    run("sad_u8", k_sad_u8);
    run("msad_u8", k_msad_u8);
    run("bitop3", k_bitop3);
    run("and_or", k_and_or);
    run("lshl_or", k_lshl_or);
    run("xad", k_xad);
    run("bfe_u", k_bfe_u);
    run("ffbl", k_ffbl);
    run("ffbh", k_ffbh);
    run("bcnt", k_bcnt);
    run("mbcnt_lo", k_mbcnt_lo);
    run("not", k_not);
    run("xor", k_xor);
    run("pk_min_u16", k_pk_min_u16);
    run("pk_min_i16", k_pk_min_i16);
    run("min3_u16", k_min3_u16);
    run("max_u16", k_max_u16);
    run("lshrrev_lit", k_lshrrev_lit);
    run("mul_u24", k_mul_u24);
    run("mul_lo", k_mul_lo);
    run("cvt_pk_u8", k_cvt_pk_u8);
    run("sdwa_sub_b", k_sdwa_sub_b);
    return 0;
}

// Micro-benchmark 5 (round 5): issue cost of the packed-f16 three-input min / max of gfx950 (v_pk_minimum3_f16 / v_pk_maximum3_f16,
// with and without op_sel half swaps) next to v_min3_i32, and of the 16-bit helpers K2's packed exact score would use; plus what a
// d16_hi LDS byte load does to the OTHER half of its destination on this part (sramecc: zeroed or preserved?).  Same harness as valu_rate4.
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
KERNEL(k_min3_i32, "v_min3_i32 %0, %0, %1, %2")
KERNEL(k_pk_max3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
KERNEL(k_pk_max3_f16_sw, "v_pk_maximum3_f16 %0, %0, %1, %2 op_sel:[0,1,1] op_sel_hi:[1,0,0]")
KERNEL(k_pk_min3_f16, "v_pk_minimum3_f16 %0, %0, %1, %2")
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
KERNEL(k_max3_f16, "v_max3_f16 %0, %0, %1, %2")
KERNEL(k_max3_f16_sel, "v_max3_f16 %0, %0, %1, %2 op_sel:[0,1,0,0]")
KERNEL(k_sub_f16, "v_sub_f16_e32 %0, %0, %1")
KERNEL(k_cvt_u16_f16, "v_cvt_u16_f16_e32 %0, %0")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x1e")
KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(k_minimum3_f32, "v_minimum3_f32 %0, %0, %1, %2")
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
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
// d16 semantics + a functional check of the packed three-input ops with half swaps
__global__ void k_d16(uint32_t* out)
{
    __shared__ uint8_t lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + threadIdx.x;
    uint32_t r = 0xAAAA5555u, r2 = 0xAAAA5555u;
    asm volatile("ds_read_u8_d16_hi %0, %2 offset:1\n\tds_read_u8_d16 %1, %2 offset:2\n\ts_waitcnt lgkmcnt(0)" : "+v"(r), "+v"(r2) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = r;           // preserved: 0x00bb5555, zeroed: 0x00bb0000
    out[threadIdx.x * 4 + 1] = r2;          // preserved: 0xAAAA00bb, zeroed: 0x000000bb
    // packed ops: a = (1024 + 5, 1024 + 9), b = (1024 + 7, 1024 + 1), c = (-(1024 + 2), 1024 + 30)
    const uint32_t a = 0x64096405u, b = 0x64016407u, c = 0x641EE402u;
    uint32_t m, s;
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));                                   // (max(5,7,-2)=7, max(9,1,30)=30) -> 0x641E6407
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(s) : "v"(a), "v"(b), "v"(c));   // c swapped: (max(5,7,30), max(9,1,-2)) -> 0x6409641E
    out[threadIdx.x * 4 + 2] = m;
    out[threadIdx.x * 4 + 3] = s;
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 64 * 16);
    k_d16<<<1, 64>>>(d);
    uint32_t h[8]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("lane 0: d16_hi into 0xAAAA5555 -> %08x   d16 into 0xAAAA5555 -> %08x   (bytes 0x0a / 0x11)\n", h[0], h[1]);
    printf("lane 0: pk_maximum3 %08x (want 641e6407)   with the third source's halves swapped %08x (want 6409641e)\n", h[2], h[3]);
    run("add", k_add);
    run("min3_i32", k_min3_i32);
    run("pk_maximum3_f16", k_pk_max3_f16);
    run("pk_maximum3_f16 sw", k_pk_max3_f16_sw);
    run("pk_minimum3_f16", k_pk_min3_f16);
    run("pk_max_f16", k_pk_max_f16);
    run("pk_add_f16", k_pk_add_f16);
    run("max3_f16", k_max3_f16);
    run("max3_f16 op_sel", k_max3_f16_sel);
    run("sub_f16", k_sub_f16);
    run("cvt_u16_f16", k_cvt_u16_f16);
    run("bitop3", k_bitop3);
    run("mad_i32_i24", k_mad_i24);
    run("minimum3_f32", k_minimum3_f32);
    run("pk_min_u16", k_pk_min_u16);
    return 0;
}

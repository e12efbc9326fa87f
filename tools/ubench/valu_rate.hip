// Micro-benchmark: issue rate of common integer VALU ops on gfx950 (wave64).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
template <int OP>
__global__ void k(uint32_t* out, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * (i + 1);
    uint32_t b = seed ^ 0x5bd1e995, c = threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = min((int)a[i], (int)b) + 0;                 // v_min_i32
            if (OP == 1) a[i] = (a[i] >> 8) & 0xFF;                         // v_bfe_u32
            if (OP == 2) a[i] = a[i] * b;                                   // v_mul_lo_u32
            if (OP == 3) a[i] = __builtin_amdgcn_alignbyte(a[i], b, c);     // v_alignbyte
            if (OP == 4) a[i] = min(min((int)a[i], (int)b), (int)c);        // v_min3_i32
            if (OP == 5) a[i] = __builtin_amdgcn_udot4(a[i], b, c, false);  // v_dot4_u32_u8
            if (OP == 6) a[i] = a[i] + b;                                   // v_add_u32
            if (OP == 7) a[i] = __popc(a[i] ^ b) + c;                       // xor + bcnt
            if (OP == 8) a[i] = (a[i] < b) ? c : a[i];                      // cmp + cndmask
        }
        b += 3; c ^= b;
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int instr_per_op)
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * 8, 256>>>(d, 1);
    hipEventRecord(e0);
    k<OP><<<256 * 8, 256>>>(d, 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = 256.0 * 8 * 4, instr = waves * ITER * 8 * instr_per_op;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s %8.3f ms  -> %.2f cycles per wave-instr per SIMD (at 2.4 GHz, %d instr/op)\n", name, ms,
           cyc / (instr / 1024.0), instr_per_op);
    hipFree(d);
}
int main()
{
    run<0>("v_min_i32", 1); run<1>("v_bfe_u32", 1); run<2>("v_mul_lo_u32", 1); run<3>("v_alignbyte", 1);
    run<4>("v_min3_i32", 1); run<5>("v_dot4_u32_u8", 1); run<6>("v_add_u32", 1); run<7>("xor+bcnt", 2);
    run<8>("cmp+cndmask", 2);
    return 0;
}

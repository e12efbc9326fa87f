// Micro-benchmark: LDS gather cost on gfx950 -- ds_read_u8 vs ds_read_b32 with per-lane scattered addresses
// (what K2's exact-score round does: 17 ring bytes per candidate), 4 waves per workgroup, 7 workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 256
template <int MODE>   // 0: u8 scattered, 1: b32 scattered, 2: u8 consecutive lanes, 3: u8 scattered, half the lanes, 4: u16 scattered
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed)
{
    extern __shared__ uint8_t sm[];
    for (int i = threadIdx.x; i < 22528 / 4; i += 256) ((uint32_t*)sm)[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t h = (threadIdx.x * 2654435761u + seed) >> 8;
    uint32_t acc = 0;
    for (int it = 0; it < ITER; it++) {
        uint32_t a = (MODE == 2) ? ((it * 37 + threadIdx.x) % 9000) : (h % 9000) + 600;
        if (MODE == 3 && (threadIdx.x & 1)) { h = h * 1664525u + 1013904223u; continue; }
        if (MODE == 1) {
            const uint32_t* p = (const uint32_t*)(sm + (a & ~3u));
            acc += p[0] + p[36] + p[-36] + p[72] + p[-72] + p[108] + p[-108] + p[1] + p[-1] + p[37] + p[-37] + p[73] + p[-73] + p[109] + p[-109] + p[2] + p[-2];
        } else if (MODE == 4) {
            const uint16_t* p = (const uint16_t*)(sm + (a & ~1u));
            acc += p[0] + p[72] + p[-72] + p[144] + p[-144] + p[216] + p[-216] + p[1] + p[-1] + p[73] + p[-73] + p[145] + p[-145] + p[217] + p[-217] + p[2] + p[-2];
        } else {
            const uint8_t* p = sm + a;
            acc += p[0] + p[144] + p[-144] + p[288] + p[-288] + p[432] + p[-432] + p[3] + p[-3] + p[145] + p[-145] + p[290] + p[-290] + p[435] + p[-435] + p[1] + p[-1];
        }
        h = h * 1664525u + 1013904223u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name)
{
    uint32_t* d; (void)hipMalloc(&d, 256 * 7 * 8 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256 * 7 * 8, 256, 22528>>>(d, 1);
    (void)hipEventRecord(e0);
    k<MODE><<<256 * 7 * 8, 256, 22528>>>(d, 2);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_cu = 7.0 * 8 * 4 * ITER * 17;     // wave-level DS instructions per CU
    printf("%-34s %7.3f ms -> %.1f cycles per DS wave-instruction per CU (@2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / inst_per_cu);
    (void)hipFree(d);
}
int main()
{
    run<0>("ds_read_u8, scattered"); run<1>("ds_read_b32, scattered"); run<2>("ds_read_u8, consecutive lanes");
    run<3>("ds_read_u8, scattered, 32 lanes"); run<4>("ds_read_u16, scattered");
    return 0;
}

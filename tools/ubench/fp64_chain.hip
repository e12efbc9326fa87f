// fp64_chain.hip -- what a running fp64 sum costs per step on gfx950 when its operands come from LDS
// (the phase-S loops of pilotguru_amd/csrc/calib.hip), one wave, 64-step chunks.
//   mode 0: v += c[i]                       operands read from LDS one batch (16 steps) ahead
//   mode 1: same, operands re-read from the same 16 LDS rows (no new addresses)
//   mode 2: v += c[i]; m = dt*v; t += m'    (the forward slot of calib.hip)
//   mode 3: mode 0 without the LDS reads (operands in registers, loaded once)
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ROWS 96
#define BATCH 16
#define REPS 2000

template <int MODE, int ACTIVE>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, double seed)
{
    __shared__ double sc[ROWS][4];
    const int lane = threadIdx.x, comp = lane % 3;
    for (int i = lane; i < ROWS * 4; i += 64) (&sc[0][0])[i] = seed + i * 1e-3;
    __syncthreads();
    double v = seed, t = 0, m1 = 0, m2 = 0, dtp = 0;
    double A[BATCH], B[BATCH], dA[BATCH], dB[BATCH];
    for (int j = 0; j < BATCH; j++) { A[j] = sc[j][comp]; B[j] = sc[16 + j][comp]; dA[j] = sc[j][3]; dB[j] = sc[16 + j][3]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (lane < ACTIVE)
    for (int rep = 0; rep < REPS; rep++) {
#pragma unroll
        for (int b = 0; b < 4; b += 2) {
            if (MODE != 3) {
#pragma unroll
                for (int j = 0; j < BATCH; j++) { A[j] = sc[(MODE == 1 ? 0 : 16 * b + 16) % 64 + j][comp]; if (MODE == 2) dA[j] = sc[(16 * b + 16) % 64 + j][3]; }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < BATCH; j++) {
                if (MODE == 2) { t += m2; m2 = m1; m1 = dtp * v; dtp = dB[j]; }
                v = v + B[j];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE != 3) {
#pragma unroll
                for (int j = 0; j < BATCH; j++) { B[j] = sc[(MODE == 1 ? 16 : 16 * b + 32) % 64 + j][comp]; if (MODE == 2) dB[j] = sc[(16 * b + 32) % 64 + j][3]; }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < BATCH; j++) {
                if (MODE == 2) { t += m2; m2 = m1; m1 = dtp * v; dtp = dA[j]; }
                v = v + A[j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[lane] = v + t + m1 + m2;
    if (lane == 0) cyc[0] = t1 - t0;
}

template <int MODE, int ACTIVE>
void run(const char* name)
{
    double* o; unsigned long long* c; unsigned long long h = 0;
    (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&c, 8);
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL((k<MODE, ACTIVE>), dim3(1), dim3(64), 0, 0, o, c, 1.0); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-60s %.2f cycles per step\n", name, (double)h / ((double)REPS * 64));
    (void)hipFree(o); (void)hipFree(c);
}

int main()
{
    run<3, 64>("v += c[i], operands in registers");
    run<3, 16>("v += c[i], operands in registers, 16 lanes active");
    run<0, 64>("v += c[i], operands streamed from LDS a batch ahead");
    run<0, 32>("the same, 32 lanes active");
    run<0, 16>("the same, 16 lanes active");
    run<0, 3>("the same, 3 lanes active");
    run<2, 64>("v += c[i]; m = dt*v; t += m'' (forward slot), LDS a batch ahead");
    run<2, 16>("the same, 16 lanes active");
    run<2, 3>("the same, 3 lanes active");
    return 0;
}

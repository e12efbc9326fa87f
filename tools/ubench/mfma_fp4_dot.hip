// Check v_mfma_scale_f32_16x16x128_f8f6f4 with fp4 (e2m1) operands +1 / -1 and E8M0 scales 2^6 * 2^6:
// D[i][j] = C + 4096 * sum_k A[i][k] * B[j][k], exact in f32.  Operand layout assumed: lane l holds
// row (column) l & 15 and 32 consecutive K values of block l >> 4 (any K order works as long as A
// and B agree).  Build: hipcc --offload-arch=gfx950 -O2 -o mfma_fp4_dot mfma_fp4_dot.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t* a, const uint32_t* b, float* d)
{
    const int l = threadIdx.x;
    v8i A = {(int)a[4 * l], (int)a[4 * l + 1], (int)a[4 * l + 2], (int)a[4 * l + 3], 0, 0, 0, 0};
    v8i B = {(int)b[4 * l], (int)b[4 * l + 1], (int)b[4 * l + 2], (int)b[4 * l + 3], 0, 0, 0, 0};
    v4f C = {1000.f + l, 1000.f + l, 1000.f + l, 1000.f + l};
    const int sc = 0x85858585;                       // E8M0 133 = 2^6
    v4f D = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, C, 4, 4, 0, sc, 0, sc);
    for (int r = 0; r < 4; r++) d[4 * l + r] = D[r];
}
int main()
{
    static int A[16][128], B[16][128];
    srand(3);
    for (int i = 0; i < 16; i++) for (int kk = 0; kk < 128; kk++) { A[i][kk] = (rand() & 1) ? 1 : -1; B[i][kk] = (rand() & 1) ? 1 : -1; }
    uint32_t ha[256] = {0}, hb[256] = {0};
    for (int l = 0; l < 64; l++)
        for (int e = 0; e < 32; e++) {
            const int kk = (l >> 4) * 32 + e, row = l & 15;
            const uint32_t na = A[row][kk] > 0 ? 0x2 : 0xA, nb = B[row][kk] > 0 ? 0x2 : 0xA;   // e2m1 +1.0 / -1.0
            ha[4 * l + e / 8] |= na << (4 * (e % 8));
            hb[4 * l + e / 8] |= nb << (4 * (e % 8));
        }
    uint32_t *da, *db; float* dd;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, da, db, dd);
    float hd[256];
    if (hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int i = 4 * (l >> 4) + r, j = l & 15;
            int dot = 0;
            for (int kk = 0; kk < 128; kk++) dot += A[i][kk] * B[j][kk];
            const float exp = 1000.f + l + 4096.f * dot;
            if (hd[4 * l + r] != exp) { if (bad < 8) printf("lane %d reg %d: got %.1f expected %.1f\n", l, r, hd[4 * l + r], exp); bad++; }
        }
    printf("%s (%d mismatches)\n", bad ? "MISMATCH" : "ok: D = C + 4096 * dot, row 4*(l>>4)+r, col l&15", bad);
    return bad ? 1 : 0;
}

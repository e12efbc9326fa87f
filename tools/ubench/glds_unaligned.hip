// Does global_load_lds (LDS-DMA) accept byte-unaligned per-lane global addresses on gfx950?
// Build: hipcc --offload-arch=gfx950 -O2 -o glds_unaligned glds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int SZ>
__global__ void k(const uint8_t* src, int shift, int pitch, uint8_t* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 16];
    const int lane = threadIdx.x;
    // lane -> row lane/4, chunk lane%4 of a row (4 chunks of SZ bytes per row)
    const uint8_t* g = src + (lane >> 2) * pitch + shift + (lane & 3) * SZ;
    if (SZ == 4) __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds, 4, 0, 0);
    else __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = 0; i < SZ; i++) out[lane * SZ + i] = lds[lane * SZ + i];
}

int main()
{
    const int pitch = 257, rows = 16;
    std::vector<uint8_t> h(pitch * rows + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 131 + (i >> 8) * 7);
    uint8_t *d, *o;
    hipMalloc(&d, h.size()); hipMalloc(&o, 64 * 16);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    int bad = 0;
    for (int sz : {4, 16})
        for (int shift = 0; shift < 8; shift++) {
            hipMemset(o, 0, 64 * 16);
            if (sz == 4) hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, d, shift, pitch, o);
            else hipLaunchKernelGGL(k<16>, 1, 64, 0, 0, d, shift, pitch, o);
            std::vector<uint8_t> r(64 * 16);
            hipMemcpy(r.data(), o, 64 * 16, hipMemcpyDeviceToHost);
            int err = 0;
            for (int lane = 0; lane < 64; lane++)
                for (int i = 0; i < sz; i++)
                    if (r[lane * sz + i] != h[(lane >> 2) * pitch + shift + (lane & 3) * sz + i]) err++;
            printf("size %2d shift %d (pitch %d): %s (%d byte errors)\n", sz, shift, pitch, err ? "MISMATCH" : "ok", err);
            bad += err;
        }
    return bad ? 1 : 0;
}

// fast.hip -- K2: per-cell FAST-9/16 score + 3x3 NMS + per-cell threshold fallback (gfx950).
//
// Restates the cell loop of ORBextractor::ComputeKeyPointsOctTree
// (thirdparty/orb-slam2/src/ORBextractor.cc:765-829), i.e. thousands of small
// cv::FAST(window, iniThFAST, true) calls with a cv::FAST(window, minThFAST, true) retry for
// empty cells (:808-816), as ONE launch: one 64-lane wave per 30-px cell, all cells of all
// pyramid levels of all frames of the batch in one grid.
//
// Equivalence used: for a pixel that is a FAST corner at threshold
// t its OpenCV score S-1 (S = best 9-arc minimum |difference|) does not depend on t, and
// "corner at t" <=> score >= t.  So the wave computes the score map of the cell interior once
// (at minThFAST), runs the 3x3 strict NMS on it with everything outside the cell interior
// counted as 0 -- exactly what the reference's per-window FAST sees -- and then emits the
// survivors with score >= iniThFAST if there are any, else those with score >= minThFAST.
//
// Structure per wave: (1) the (wCell+6)x(hCell+6) window is staged into LDS with aligned
// 32-bit global loads (coalesced rows, ~1.4x halo re-read served by L2); (2) a cheap
// necessary test (two opposite ring pairs) compacts the few percent of plausible pixels into
// an LDS list with ballot/mbcnt; (3) full 16-ring scores are computed for the compacted list
// with all lanes busy; (4) NMS + per-cell threshold + one atomicAdd per cell to reserve
// slots in the (frame, level) candidate array.  Candidate order in that array is arbitrary;
// everything downstream orders by the reference's (cell row, cell col, y, x) rank.
//
// Roofline: HBM/L2-read bound in principle (one pass over all pyramid pixels, 1 B/px);
// algorithmic bytes per frame = sum_l w_l*h_l.
#include "pgorb_internal.h"

extern __shared__ __attribute__((aligned(16))) uint8_t pg_fast_smem[];

__device__ __forceinline__ int imin3(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }

// 16-ring offsets in OpenCV's order (x, y): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)
// (0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
__device__ __forceinline__ void ring_load(const uint8_t* c, int p, int v, int d[16])
{
    d[0] = v - c[3 * p];          d[1] = v - c[3 * p + 1];    d[2] = v - c[2 * p + 2];
    d[3] = v - c[p + 3];          d[4] = v - c[3];            d[5] = v - c[-p + 3];
    d[6] = v - c[-2 * p + 2];     d[7] = v - c[-3 * p + 1];   d[8] = v - c[-3 * p];
    d[9] = v - c[-3 * p - 1];     d[10] = v - c[-2 * p - 2];  d[11] = v - c[-p - 3];
    d[12] = v - c[-3];            d[13] = v - c[p - 3];       d[14] = v - c[2 * p - 2];
    d[15] = v - c[3 * p - 1];
}

// OpenCV cornerScore<16> for a corner: max over the 16 nine-long arcs of the arc minimum of
// d (darker ring) or of -d (brighter ring), minus 1.
__device__ __forceinline__ int fast_score16(const int d[16])
{
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = imin3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        hi3[k] = imax3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    }
    int best_dark = -1000, best_bright = 1000;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        best_dark = max(best_dark, imin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]));
        best_bright = min(best_bright, imax3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]));
    }
    return max(best_dark, -best_bright) - 1;
}

__global__ __launch_bounds__(64) void k_fast_cells(const PgPlan P, int tilePitch, int tileRows,
                                                    int mapPitch, int mapRows)
{
    const int lane = threadIdx.x;
    const int frame = blockIdx.y;
    int l = 0;
    while (l + 1 < P.nlevels && (int)blockIdx.x >= P.lvl[l + 1].cellBase) l++;
    const PgLevel& L = P.lvl[l];
    const int c = blockIdx.x - L.cellBase;
    const int ci = c / L.nCols, cj = c - ci * L.nCols;
    const int maxBorderX = L.w - PG_EDGE, maxBorderY = L.h - PG_EDGE;
    const int iniY = PG_EDGE + ci * L.hCell;
    const int iniX = PG_EDGE + cj * L.wCell;
    if (iniY >= maxBorderY - 3 || iniX >= maxBorderX - 6) return;      // :794, :803
    const int maxX = min(iniX + L.wCell + 6, maxBorderX);
    const int maxY = min(iniY + L.hCell + 6, maxBorderY);
    const int W = maxX - iniX, H = maxY - iniY;
    if (W < 7 || H < 7) return;                                         // cv::FAST finds nothing
    const int IW = W - 6, IH = H - 6;

    uint8_t* tile = pg_fast_smem;                                  // [tileRows][tilePitch]
    uint8_t* smap = tile + tileRows * tilePitch;                   // [mapRows][mapPitch], 1-px zero rim
    uint16_t* list = reinterpret_cast<uint16_t*>(smap + mapRows * mapPitch);   // pixel ids
    uint8_t* lscore = reinterpret_cast<uint8_t*>(list + (mapRows - 2) * (mapPitch - 2 > 0 ? mapPitch : 1));

    // (1) stage the window: aligned dwords covering [iniX, maxX) of rows [iniY, maxY)
    const uint8_t* img = L.img + (int64_t)frame * L.fstride;
    const int xa = iniX & ~3, shift = iniX - xa;
    const int ndw = (shift + W + 3) >> 2;
    for (int i = lane; i < ndw * H; i += 64) {
        const int r = i / ndw, q = i - r * ndw;
        const uint32_t v = *reinterpret_cast<const uint32_t*>(img + (int64_t)(iniY + r) * L.pitch + xa + 4 * q);
        *reinterpret_cast<uint32_t*>(tile + r * tilePitch + 4 * q) = v;
    }
    for (int i = lane; i < (mapRows * mapPitch) >> 2; i += 64)
        reinterpret_cast<uint32_t*>(smap)[i] = 0;
    __syncthreads();

    // (2) necessary test + compaction.  Any 9-arc of the 16-ring contains at least one pixel
    // of every opposite pair {k, k+8}; test pairs (0,8) and (4,12).
    const int t = P.minTh;
    int nlist = 0;
    const int npix = IW * IH;
    for (int base = 0; base < npix; base += 64) {
        const int p = base + lane;
        bool pass = false;
        if (p < npix) {
            const int iy = p / IW, ix = p - iy * IW;
            const uint8_t* cp = tile + (iy + 3) * tilePitch + shift + ix + 3;
            const int v = cp[0];
            const int r0 = cp[3 * tilePitch], r8 = cp[-3 * tilePitch], r4 = cp[3], r12 = cp[-3];
            const int lo = v - t, hi = v + t;
            const bool dark = (r0 < lo || r8 < lo) && (r4 < lo || r12 < lo);
            const bool bright = (r0 > hi || r8 > hi) && (r4 > hi || r12 > hi);
            pass = dark || bright;
        }
        const unsigned long long m = __ballot(pass);
        if (pass) {
            const int pos = nlist + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                       __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            list[pos] = (uint16_t)p;
        }
        nlist += __popcll(m);
    }
    __syncthreads();

    // (3) exact scores for the compacted pixels
    for (int base = 0; base < nlist; base += 64) {
        const int i = base + lane;
        if (i < nlist) {
            const int p = list[i];
            const int iy = p / IW, ix = p - iy * IW;
            const uint8_t* cp = tile + (iy + 3) * tilePitch + shift + ix + 3;
            int d[16];
            ring_load(cp, tilePitch, cp[0], d);
            int s = fast_score16(d);
            s = (s >= t) ? s : 0;                    // not a corner at minThFAST
            lscore[i] = (uint8_t)s;
            if (s) smap[(iy + 1) * mapPitch + ix + 1] = (uint8_t)s;
        }
    }
    __syncthreads();

    // (4) NMS (strictly greater than all 8 neighbours; outside the interior = 0), per-cell
    // threshold choice, slot reservation, emit.
    bool anyIni = false;
    int nsurv = 0;
    for (int base = 0; base < nlist; base += 64) {
        const int i = base + lane;
        bool keep = false;
        int s = 0;
        if (i < nlist) {
            s = lscore[i];
            if (s) {
                const int p = list[i];
                const int iy = p / IW, ix = p - iy * IW;
                const uint8_t* m = smap + (iy + 1) * mapPitch + ix + 1;
                keep = s > m[-1] && s > m[1] && s > m[-mapPitch - 1] && s > m[-mapPitch] &&
                       s > m[-mapPitch + 1] && s > m[mapPitch - 1] && s > m[mapPitch] &&
                       s > m[mapPitch + 1];
            }
            lscore[i] = keep ? (uint8_t)s : 0;       // reuse as "survivor score"
        }
        anyIni |= (__ballot(keep && s >= P.iniTh) != 0ull);
        nsurv += __popcll(__ballot(keep));
    }
    if (nsurv == 0) return;
    __syncthreads();
    const int thr = anyIni ? P.iniTh : t;

    int total = 0;
    for (int base = 0; base < nlist; base += 64) {
        const int i = base + lane;
        const bool emit = (i < nlist) && lscore[i] >= thr && lscore[i] != 0;
        total += __popcll(__ballot(emit));
    }
    int slot0 = 0;
    if (lane == 0) slot0 = atomicAdd(&P.candCount[frame * PG_MAXL + l], total);
    slot0 = __shfl(slot0, 0);
    uint32_t* out = P.cand + (int64_t)frame * P.candFrame + L.candOff;
    int done = 0;
    for (int base = 0; base < nlist; base += 64) {
        const int i = base + lane;
        const bool emit = (i < nlist) && lscore[i] >= thr && lscore[i] != 0;
        const unsigned long long m = __ballot(emit);
        if (emit) {
            const int pos = slot0 + done + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                              __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            const int p = list[i];
            const int iy = p / IW, ix = p - iy * IW;
            // region-relative coordinates: window-local + cell offset (:822-823)
            const int xr = ix + 3 + cj * L.wCell, yr = iy + 3 + ci * L.hCell;
            if (pos < L.candCap)
                out[pos] = (uint32_t)xr | ((uint32_t)yr << 12) | ((uint32_t)lscore[i] << 24);
            else
                atomicExch(P.status, PGORB_E_OVERFLOW);
        }
        done += __popcll(m);
    }
}

void pg_launch_fast(const PgPlan& P, int nframes, hipStream_t s)
{
    int maxW = 0, maxH = 0;
    for (int l = 0; l < P.nlevels; l++) {
        maxW = max(maxW, P.lvl[l].wCell + 6);
        maxH = max(maxH, P.lvl[l].hCell + 6);
    }
    const int tilePitch = ((maxW + 3 + 3) & ~3) + 4;       // shift<=3, round up to dwords, +4 pad
    const int tileRows = maxH;
    const int mapPitch = ((maxW - 6 + 2) + 3) & ~3;
    const int mapRows = maxH - 6 + 2;
    const int npixMax = (mapRows - 2) * mapPitch;
    size_t smem = (size_t)tileRows * tilePitch + (size_t)mapRows * mapPitch;
    smem = (smem + 15) & ~(size_t)15;
    smem += (size_t)npixMax * 2 + (size_t)npixMax;
    dim3 grid(P.totalCells, nframes), block(64);
    hipLaunchKernelGGL(k_fast_cells, grid, block, smem, s, P, tilePitch, tileRows, mapPitch, mapRows);
}

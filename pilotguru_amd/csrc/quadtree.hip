// quadtree.hip -- K3: ORBextractor::DistributeOctTree as a generation-synchronous kernel.
//
// Restates thirdparty/orb-slam2/src/ORBextractor.cc:539-763 (DistributeOctTree) and :481-537
// (ExtractorNode::DivideNode).  The reference is a sequential std::list algorithm; here one
// 1024-thread workgroup owns one (frame, level) problem and advances it one *generation* at a
// time, every step inside a generation being a data-parallel pass over keys or nodes:
//
//   * keys never move.  Each key carries the list position of the node that holds it; the
//     "first maximum in vKeys order wins" rule (:747-757) only needs the reference's
//     candidate order, which is a closed-form rank of (x, y) (cell row, cell col, y, x).
//   * the list is an array in list order, resident in LDS.  One pass of the reference's outer
//     loop (:606-665) splits every expandable node; because children are push_front'ed
//     (:623-660) the new list is [children of the LAST processed parent as n4,n3,n2,n1 ...
//     children of the first] ++ [untouched single-key nodes in their old order] -- a suffix sum.
//   * the reference's final "largest first" phase (:673-737) sorts (size, node*) ascending and
//     walks from the back with an early break at N; one inner iteration is again a generation
//     whose processing order is the sort order and whose processed prefix is found with a
//     prefix sum.  PARITY CONTRACT for the pointer tie (:684): equal-sized nodes are split in
//     order of creation sequence, later created first (SURVEY.md hard part 2, same rule as
//     oracle/orb_oracle.c).  All expandable nodes alive at that point were created in the
//     previous generation and sit in its head group in exact reverse creation order, so
//     "later created first" == "smaller list position first".
//   * one pass over the keys per generation: the pass that moves keys to their new node also
//     counts them into that node's quadrants for the next generation (LDS atomics).
//
// The prologue compacts K2's per-cell candidate slots (one thread per cell) into
// the dense 8-byte key records {packed candidate, node position | quadrant << 28}.
//
// Integer/compare work on ~1e4 keys; latency-bound, not bandwidth-bound.  Throughput comes
// from running (levels x frames) workgroups concurrently.
#include "pgorb_internal.h"

#define QT_T 1024
#define QT_W (QT_T / 64)
#define QT_POS_MASK 0x0FFFFFFFu

// Exclusive prefix sum of a[0..n) in place (a in LDS or global); returns the total.
// All threads must call.  sh: >= QT_W + 1 ints of LDS.
__device__ int qt_scan_excl(int* a, int n, int* sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n + QT_T - 1) / QT_T;
    const int b = tid * per, e = min(b + per, n);
    int sum = 0;
    for (int i = b; i < e; i++) sum += a[i];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        int v = (lane < QT_W) ? sh[lane] : 0, w = v;
#pragma unroll
        for (int d = 1; d < QT_W; d <<= 1) {
            const int o = __shfl_up(w, d);
            if (lane >= d) w += o;
        }
        if (lane < QT_W) sh[lane] = w - v;
        if (lane == QT_W - 1) sh[QT_W] = w;
    }
    __syncthreads();
    int run = sh[wv] + incl - sum;
    for (int i = b; i < e; i++) { const int v = a[i]; a[i] = run; run += v; }
    const int total = sh[QT_W];
    __syncthreads();
    return total;
}

__device__ __forceinline__ int qt_quadrant(const int4 b, uint32_t cv)
{
    const int midX = b.x + ((b.z - b.x + 1) >> 1);     // UL.x + ceil((UR.x-UL.x)/2)  (:483)
    const int midY = b.y + ((b.w - b.y + 1) >> 1);
    const int x = cv & 0xFFF, y = (cv >> 12) & 0xFFF;
    return (x < midX) ? ((y < midY) ? 0 : 2) : ((y < midY) ? 1 : 3);      // n1 n3 / n2 n4 (:515-526)
}

__global__ __launch_bounds__(QT_T, 8) void k_quadtree(const PgPlan P)
{
    __shared__ int sh[QT_W + 8];
    extern __shared__ __attribute__((aligned(16))) int qt_lds[];     // 24 ints per node
    const int tid = threadIdx.x;
    // grid = (frames, levels): the heavy level-0 problems of all frames are dispatched first and
    // spread over all CUs (with level as the fast index every 8th workgroup -- always the same
    // 32 CUs under round-robin dispatch -- got all the level-0 work)
    const int l = blockIdx.y, frame = blockIdx.x;
    const PgLevel& L = P.lvl[l];
    int* kpc = &P.kpCount[frame * PG_MAXL + l];
    uint2* keys = reinterpret_cast<uint2*>(P.cand) + ((int64_t)frame * P.candFrame + L.candOff);
    const int NC = L.nodeCap;
    int4* bndA = reinterpret_cast<int4*>(qt_lds);      // (ULx, ULy, URx, BRy) in list order
    int4* bndB = bndA + NC;
    int* cntA = reinterpret_cast<int*>(bndB + NC);
    int* cntB = cntA + NC;
    int* cnt4 = cntB + NC;                              // [4*NC] keys per child of node p
    int* newpos4 = cnt4 + 4 * NC;                       // [4*NC] new list position of child
    int* tailpos = newpos4 + 4 * NC;                    // [NC]  new position of unprocessed node
    int* rnk = tailpos + NC;                            // [NC]  processing rank or -1
    int* ord = rnk + NC;                                // [NC]  node at processing rank r
    int* cinc = ord + NC;                               // [NC]  inclusive sum of child counts
    int* tmp = cinc + NC;                               // [NC]
    int* ecnt = tmp + NC;                               // [NC]  sizes of expandable nodes
    unsigned long long* best = reinterpret_cast<unsigned long long*>(newpos4);    // [NC] (epilogue)

    // ---- prologue: compact K2's per-cell slots into dense key records -----------------------
    int ncand;
    {
        const int ncells = L.nCols * L.nRows;
        const int32_t* cc = P.cellCount + (int64_t)frame * P.totalCells + L.cellBase;
        const uint32_t* slots = P.cellCand + (int64_t)frame * P.cellCandFrame + L.cellCandOff;
        int* cellOff = P.nodeScratch + (int64_t)frame * P.nodeFrame + L.nodeOff;    // [ncells]
        for (int i = tid; i < ncells; i += QT_T) cellOff[i] = cc[i];
        __syncthreads();
        ncand = qt_scan_excl(cellOff, ncells, sh);
        if (ncand > L.candCap) ncand = L.candCap;               // cannot happen (capacity is exact)
        // one thread per cell (a cell holds ~10 records); loads of a cell are issued 4 at a time.
        // (A wave-per-cell loop serialised ncells/16 dependent global round trips: it WAS the kernel.)
        for (int c = tid; c < ncells; c += QT_T) {
            const int n = min(cc[c], L.cellCap), off = cellOff[c];
            const uint32_t* src = slots + (int64_t)c * L.cellCap;
            for (int j = 0; j < n; j += 4) {
                uint32_t v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = (j + u < n) ? src[j + u] : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (j + u < n && off + j + u < L.candCap) keys[off + j + u] = make_uint2(v[u], 0u);
            }
        }
        if (tid == 0) P.candCount[frame * PG_MAXL + l] = ncand;
        __syncthreads();
    }
    if (ncand <= 0) { if (tid == 0) *kpc = 0; return; }

    const int N = L.quota;
    const int regionH = L.h - 2 * PG_EDGE;              // maxY - minY
    int size;

    // ---- initial nodes (:543-585) ------------------------------------------------------
    {
        const int nIni = L.nIni;
        const float hX = L.hX;
        int* rootCnt = cnt4;
        for (int i = tid; i < nIni; i += QT_T) rootCnt[i] = 0;
        __syncthreads();
        for (int i = tid; i < ncand; i += QT_T) {
            const int x = keys[i].x & 0xFFF;
            int r = (int)__fdiv_rn((float)x, hX);       // vpIniNodes[kp.pt.x/hX]  (:569)
            r = min(max(r, 0), nIni - 1);
            keys[i].y = (uint32_t)r;
            atomicAdd(&rootCnt[r], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int pos = 0;
            for (int i = 0; i < nIni; i++) {
                if (rootCnt[i] > 0) {
                    bndA[pos] = make_int4((int)__fmul_rn(hX, (float)i), 0,
                                          (int)__fmul_rn(hX, (float)(i + 1)), regionH);
                    cntA[pos] = rootCnt[i];
                    tailpos[i] = pos++;
                } else tailpos[i] = -1;
            }
            sh[QT_W + 1] = pos;
        }
        __syncthreads();
        size = sh[QT_W + 1];
        __syncthreads();
        for (int i = tid; i < 4 * size; i += QT_T) cnt4[i] = 0;
        __syncthreads();
        // move keys to the compacted root list and count quadrants for generation 1
        for (int i = tid; i < ncand; i += QT_T) {
            const uint2 k = keys[i];
            const int pos = tailpos[k.y];
            uint32_t rec = (uint32_t)pos;
            if (cntA[pos] > 1) {
                const int q = qt_quadrant(bndA[pos], k.x);
                atomicAdd(&cnt4[pos * 4 + q], 1);
                rec |= (uint32_t)q << 28;
            }
            keys[i].y = rec;
        }
        __syncthreads();
    }

    // ---- generations (:594-739) ---------------------------------------------------------
    int sorted_mode = 0;
    bool last = false;
    while (!last) {
        const int n = size, prevSize = size;
        // processing order
        for (int p = tid; p < n; p += QT_T) rnk[p] = (cntA[p] > 1) ? 1 : 0;
        __syncthreads();
        const int m = qt_scan_excl(rnk, n, sh);          // rnk[p] = list-order rank among expandable
        for (int p = tid; p < n; p += QT_T) {
            if (cntA[p] > 1) { tmp[rnk[p]] = p; ecnt[rnk[p]] = cntA[p]; }
            else rnk[p] = -1;
        }
        __syncthreads();
        if (!sorted_mode) {
            for (int r = tid; r < m; r += QT_T) ord[r] = tmp[r];
        } else {
            // descending size, ties: smaller list position first (see header)
            for (int i = tid; i < m; i += QT_T) {
                const int ci = ecnt[i];
                int rank = 0;
                for (int j = 0; j < m; j++) {
                    const int cj = ecnt[j];
                    rank += (cj > ci) || (cj == ci && j < i);
                }
                ord[rank] = tmp[i];
                rnk[tmp[i]] = rank;
            }
        }
        __syncthreads();
        // inclusive sum of non-empty child counts in processing order
        for (int r = tid; r < m; r += QT_T) {
            const int p = ord[r];
            cinc[r] = (cnt4[4 * p] > 0) + (cnt4[4 * p + 1] > 0) + (cnt4[4 * p + 2] > 0) + (cnt4[4 * p + 3] > 0);
        }
        __syncthreads();
        qt_scan_excl(cinc, m, sh);                       // exclusive ...
        for (int r = tid; r < m; r += QT_T) {            // ... -> inclusive
            const int p = ord[r];
            cinc[r] += (cnt4[4 * p] > 0) + (cnt4[4 * p + 1] > 0) + (cnt4[4 * p + 2] > 0) + (cnt4[4 * p + 3] > 0);
        }
        if (tid == 0) { sh[QT_W + 1] = m - 1; sh[QT_W + 2] = 0; }
        __syncthreads();
        if (sorted_mode) {                               // early break at N (:730)
            for (int r = tid; r < m; r += QT_T) {
                const bool now = n + cinc[r] - (r + 1) >= N;
                const bool before = (r > 0) && (n + cinc[r - 1] - r >= N);
                if (now && !before) sh[QT_W + 1] = r;
            }
            __syncthreads();
        }
        const int jstar = sh[QT_W + 1];
        const int Ctot = (m > 0) ? cinc[jstar] : 0;
        // unprocessed nodes keep their relative order behind the new children
        for (int p = tid; p < n; p += QT_T) tailpos[p] = (rnk[p] >= 0 && rnk[p] <= jstar) ? 0 : 1;
        __syncthreads();
        const int U = qt_scan_excl(tailpos, n, sh);
        // write the new list
        int myExpand = 0;
        for (int p = tid; p < n; p += QT_T) {
            const int r = rnk[p];
            if (r >= 0 && r <= jstar) {
                const int4 b = bndA[p];
                const int midX = b.x + ((b.z - b.x + 1) >> 1);
                const int midY = b.y + ((b.w - b.y + 1) >> 1);
                const int c0 = cnt4[4 * p], c1 = cnt4[4 * p + 1], c2 = cnt4[4 * p + 2], c3 = cnt4[4 * p + 3];
                int pos = Ctot - cinc[r];                // children of later-processed parents are in front
                if (c3 > 0) { bndB[pos] = make_int4(midX, midY, b.z, b.w); cntB[pos] = c3; newpos4[4 * p + 3] = pos++; }
                if (c2 > 0) { bndB[pos] = make_int4(b.x, midY, midX, b.w); cntB[pos] = c2; newpos4[4 * p + 2] = pos++; }
                if (c1 > 0) { bndB[pos] = make_int4(midX, b.y, b.z, midY); cntB[pos] = c1; newpos4[4 * p + 1] = pos++; }
                if (c0 > 0) { bndB[pos] = make_int4(b.x, b.y, midX, midY); cntB[pos] = c0; newpos4[4 * p] = pos++; }
                myExpand += (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
            } else {
                const int pos = Ctot + tailpos[p];
                bndB[pos] = bndA[p]; cntB[pos] = cntA[p];
                tailpos[p] = pos;
            }
        }
        if (myExpand) atomicAdd(&sh[QT_W + 2], myExpand);
        __syncthreads();
        const int nToExpand = sh[QT_W + 2];
        size = Ctot + U;
        if (size > NC) { if (tid == 0) atomicExch(P.status, PGORB_E_OVERFLOW); size = NC; last = true; }
        if (size >= N || size == prevSize) last = true;                    // :669 / :734
        else if (!sorted_mode && size + 3 * nToExpand > N) sorted_mode = 1; // :673
        // keys move to their new node; unless this was the last generation they are also
        // counted into that node's quadrants (cnt4 of the next generation)
        for (int i = tid; i < 4 * size; i += QT_T) cnt4[i] = 0;
        __syncthreads();
        // 4 keys per thread and step: the four independent global loads are in flight together
        // (one key per step left this pass bound by L2 latency)
        for (int i0 = tid; i0 < ncand; i0 += 4 * QT_T) {
            uint2 kk[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + u * QT_T; kk[u] = (i < ncand) ? keys[i] : make_uint2(0u, 0u); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * QT_T;
                if (i >= ncand) break;
                const uint2 k = kk[u];
                const int pos = k.y & QT_POS_MASK, q = k.y >> 28;
                const int r = rnk[pos];
                const int np = (r >= 0 && r <= jstar) ? newpos4[4 * pos + q] : tailpos[pos];
                uint32_t rec = (uint32_t)np;
                if (!last && cntB[np] > 1) {
                    const int q2 = qt_quadrant(bndB[np], k.x);
                    atomicAdd(&cnt4[np * 4 + q2], 1);
                    rec |= (uint32_t)q2 << 28;
                }
                keys[i].y = rec;
            }
        }
        __syncthreads();
        { int4* t4 = bndA; bndA = bndB; bndB = t4; int* t1 = cntA; cntA = cntB; cntB = t1; }
    }

    // ---- best response per node, first in candidate order wins (:741-760) ---------------
    for (int p = tid; p < size; p += QT_T) best[p] = 0ull;
    __syncthreads();
    const int wCell = L.wCell, hCell = L.hCell, nCols = L.nCols;
    for (int i0 = tid; i0 < ncand; i0 += 4 * QT_T) {
        uint2 kk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + u * QT_T; kk[u] = (i < ncand) ? keys[i] : make_uint2(0u, 0u); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (i0 + u * QT_T >= ncand) break;
            const uint2 k = kk[u];
            const int x = (k.x & 0xFFF) - 3, y = ((k.x >> 12) & 0xFFF) - 3;
            const int cj = x / wCell, ci = y / hCell;
            const uint32_t rank = (uint32_t)(((ci * nCols + cj) * hCell + (y - ci * hCell)) * wCell + (x - cj * wCell));
            atomicMax(&best[k.y & QT_POS_MASK], ((unsigned long long)(k.x >> 24) << 32) | (0xFFFFFFFFu - rank));
        }
    }
    __syncthreads();
    uint32_t* sel = P.sel + (int64_t)frame * P.selFrame + L.selOff;
    const int nsel = min(size, L.selCap);
    if (size > L.selCap && tid == 0) atomicExch(P.status, PGORB_E_OVERFLOW);
    for (int i = tid; i < ncand; i += QT_T) {
        const uint2 k = keys[i];
        const int x = (k.x & 0xFFF) - 3, y = ((k.x >> 12) & 0xFFF) - 3;
        const int cj = x / wCell, ci = y / hCell;
        const uint32_t rank = (uint32_t)(((ci * nCols + cj) * hCell + (y - ci * hCell)) * wCell + (x - cj * wCell));
        const unsigned long long key = ((unsigned long long)(k.x >> 24) << 32) | (0xFFFFFFFFu - rank);
        const int pos = k.y & QT_POS_MASK;
        if (pos < nsel && best[pos] == key) sel[pos] = k.x;
    }
    if (tid == 0) *kpc = nsel;
}

void pg_launch_quadtree(const PgPlan& P, int nframes, hipStream_t s)
{
    int ncMax = 0;
    for (int l = 0; l < P.nlevels; l++) ncMax = max(ncMax, P.lvl[l].nodeCap);
    const size_t lds = (size_t)ncMax * 24 * sizeof(int);
    static size_t configured = 0;
    if (lds > configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = lds;
    }
    dim3 grid(nframes, P.nlevels), block(QT_T);
    hipLaunchKernelGGL(k_quadtree, grid, block, lds, s, P);
}

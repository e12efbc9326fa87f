// quadtree.hip -- K3: ORBextractor::DistributeOctTree as a generation-synchronous kernel.
//
// Restates thirdparty/orb-slam2/src/ORBextractor.cc:539-763 (DistributeOctTree) and :481-537
// (ExtractorNode::DivideNode).  The reference is a sequential std::list algorithm; here one
// 256-thread workgroup owns one (frame, level) problem and advances it one *generation* at a
// time, every step inside a generation being a data-parallel pass over keys or nodes:
//
//   * keys never move.  Each key carries the list position of the node that holds it; the
//     "first maximum in vKeys order wins" rule (:747-757) only needs the reference's
//     candidate order, which is a closed-form rank of (x, y) (cell row, cell col, y, x).
//   * the list is an array in list order.  One pass of the reference's outer loop (:606-665)
//     splits every expandable node; because children are push_front'ed (:623-660) the new
//     list is [children of the LAST processed parent as n4,n3,n2,n1 ... children of the
//     first] ++ [untouched single-key nodes in their old order] -- a suffix sum.
//   * the reference's final "largest first" phase (:673-737) sorts (size, node*) ascending and
//     walks from the back with an early break at N; one inner iteration is again a generation
//     whose processing order is the sort order and whose processed prefix is found with a
//     prefix sum.  PARITY CONTRACT for the pointer tie (:684): equal-sized nodes are split in
//     order of creation sequence, later created first (SURVEY.md hard part 2, same rule as
//     oracle/orb_oracle.c).  All expandable nodes alive at that point were created in the
//     previous generation and sit in its head group in exact reverse creation order, so
//     "later created first" == "smaller list position first".
//
// Integer/compare work on ~1e4 keys; latency-bound, not bandwidth-bound.  Throughput comes
// from running (levels x frames) workgroups concurrently.
#include "pgorb_internal.h"

#define QT_T 256
#define QT_POS_MASK 0x0FFFFFFFu

// Exclusive prefix sum of a[0..n) in place; returns the total.  All threads must call.
__device__ int qt_scan_excl(int* a, int n, int* sh)
{
    const int tid = threadIdx.x;
    const int per = (n + QT_T - 1) / QT_T;
    const int b = tid * per, e = min(b + per, n);
    int sum = 0;
    for (int i = b; i < e; i++) sum += a[i];
    sh[tid] = sum;
    __syncthreads();
    if (tid < 64) {                      // wave 0 scans the 256 partials, 4 per lane
        int v0 = sh[4 * tid], v1 = sh[4 * tid + 1], v2 = sh[4 * tid + 2], v3 = sh[4 * tid + 3];
        int s = v0 + v1 + v2 + v3, incl = s;
        for (int d = 1; d < 64; d <<= 1) {
            int o = __shfl_up(incl, d);
            if (tid >= d) incl += o;
        }
        int ex = incl - s;
        sh[4 * tid] = ex; sh[4 * tid + 1] = ex + v0; sh[4 * tid + 2] = ex + v0 + v1;
        sh[4 * tid + 3] = ex + v0 + v1 + v2;
        if (tid == 63) sh[QT_T] = incl;
    }
    __syncthreads();
    int run = sh[tid];
    for (int i = b; i < e; i++) { int v = a[i]; a[i] = run; run += v; }
    const int total = sh[QT_T];
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(QT_T) void k_quadtree(const PgPlan P)
{
    __shared__ int sh[QT_T + 8];
    extern __shared__ __attribute__((aligned(16))) int qt_lds[];     // 6*nodeCap ints
    const int tid = threadIdx.x;
    const int l = blockIdx.x, frame = blockIdx.y;
    const PgLevel& L = P.lvl[l];
    int* kpc = &P.kpCount[frame * PG_MAXL + l];
    uint32_t* cand = P.cand + (int64_t)frame * P.candFrame + L.candOff;

    // ---- prologue: compact K2's per-cell slots into a dense candidate array ---------------
    int ncand;
    {
        const int ncells = L.nCols * L.nRows;
        const int32_t* cc = P.cellCount + (int64_t)frame * P.totalCells + L.cellBase;
        const uint32_t* slots = P.cellCand + (int64_t)frame * P.cellCandFrame + L.cellCandOff;
        const int per = (ncells + QT_T - 1) / QT_T;
        const int cb = tid * per, ce = min(cb + per, ncells);
        int sum = 0;
        for (int i = cb; i < ce; i++) sum += cc[i];
        sh[tid] = sum;
        __syncthreads();
        if (tid < 64) {
            int v0 = sh[4 * tid], v1 = sh[4 * tid + 1], v2 = sh[4 * tid + 2], v3 = sh[4 * tid + 3];
            int s = v0 + v1 + v2 + v3, incl = s;
            for (int d = 1; d < 64; d <<= 1) {
                int o = __shfl_up(incl, d);
                if (tid >= d) incl += o;
            }
            int ex = incl - s;
            sh[4 * tid] = ex; sh[4 * tid + 1] = ex + v0; sh[4 * tid + 2] = ex + v0 + v1;
            sh[4 * tid + 3] = ex + v0 + v1 + v2;
            if (tid == 63) sh[QT_T] = incl;
        }
        __syncthreads();
        int run = sh[tid];
        ncand = sh[QT_T];
        for (int i = cb; i < ce; i++) {
            const int n = cc[i];
            const uint32_t* src = slots + (int64_t)i * L.cellCap;
            for (int j = 0; j < n; j++) cand[run + j] = src[j];
            run += n;
        }
        if (tid == 0) P.candCount[frame * PG_MAXL + l] = ncand;
        __syncthreads();
    }
    if (ncand > L.candCap) ncand = L.candCap;
    if (ncand <= 0) { if (tid == 0) *kpc = 0; return; }

    uint32_t* kpos = P.kpos + (int64_t)frame * P.candFrame + L.candOff;
    int* S = P.nodeScratch + (int64_t)frame * P.nodeFrame + L.nodeOff;
    const int NC = L.nodeCap;
    int4* bndA = reinterpret_cast<int4*>(S);            // (ULx, ULy, URx, BRy) in list order
    int4* bndB = bndA + NC;
    int* cntA = reinterpret_cast<int*>(bndB + NC);
    int* cntB = cntA + NC;
    int* newpos4 = cntB + NC;                           // [4*NC] new list position of child
    int* tailpos = newpos4 + 4 * NC;                    // [NC]  new position of unprocessed node
    int* rnk = tailpos + NC;                            // [NC]  processing rank or -1
    int* ord = rnk + NC;                                // [NC]  node at processing rank r
    int* cinc = ord + NC;                               // [NC]  inclusive sum of child counts
    int* tmp = cinc + NC;                               // [NC]
    // Everything that is updated with atomics lives in LDS: global atomics execute in L2 and
    // a later plain load could hit a stale line in this CU's L1.
    int* cnt4 = qt_lds;                                 // [4*NC] keys per child of node p
    int* ecnt = qt_lds + 4 * NC;                        // [NC]   sizes of expandable nodes
    unsigned long long* best = reinterpret_cast<unsigned long long*>(qt_lds);      // [NC] (epilogue)

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
            const int x = cand[i] & 0xFFF;
            int r = (int)__fdiv_rn((float)x, hX);       // vpIniNodes[kp.pt.x/hX]  (:569)
            r = min(max(r, 0), nIni - 1);
            kpos[i] = (uint32_t)r;
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
            sh[0] = pos;
        }
        __syncthreads();
        size = sh[0];
        __syncthreads();
        for (int i = tid; i < ncand; i += QT_T) kpos[i] = (uint32_t)tailpos[kpos[i]];
        __syncthreads();
    }

    // ---- generations (:594-739) ---------------------------------------------------------
    int sorted_mode = 0;
    for (;;) {
        const int n = size, prevSize = size;
        for (int i = tid; i < 4 * n; i += QT_T) cnt4[i] = 0;
        __syncthreads();
        // key -> child quadrant (DivideNode :483-526)
        for (int i = tid; i < ncand; i += QT_T) {
            const int pos = kpos[i] & QT_POS_MASK;
            if (cntA[pos] > 1) {
                const int4 b = bndA[pos];
                const int midX = b.x + ((b.z - b.x + 1) >> 1);     // UL.x + ceil((UR.x-UL.x)/2)
                const int midY = b.y + ((b.w - b.y + 1) >> 1);
                const uint32_t cv = cand[i];
                const int x = cv & 0xFFF, y = (cv >> 12) & 0xFFF;
                const int q = (x < midX) ? ((y < midY) ? 0 : 2) : ((y < midY) ? 1 : 3);
                atomicAdd(&cnt4[pos * 4 + q], 1);
                kpos[i] = (uint32_t)pos | ((uint32_t)q << 28);
            }
        }
        __syncthreads();
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
        if (tid == 0) sh[QT_T + 1] = m - 1;
        __syncthreads();
        if (sorted_mode) {                               // early break at N (:730)
            for (int r = tid; r < m; r += QT_T) {
                const bool now = n + cinc[r] - (r + 1) >= N;
                const bool before = (r > 0) && (n + cinc[r - 1] - r >= N);
                if (now && !before) sh[QT_T + 1] = r;
            }
            __syncthreads();
        }
        const int jstar = sh[QT_T + 1];
        const int Ctot = (m > 0) ? cinc[jstar] : 0;
        // unprocessed nodes keep their relative order behind the new children
        for (int p = tid; p < n; p += QT_T) tailpos[p] = (rnk[p] >= 0 && rnk[p] <= jstar) ? 0 : 1;
        if (tid == 0) sh[QT_T + 2] = 0;
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
        if (myExpand) atomicAdd(&sh[QT_T + 2], myExpand);
        __syncthreads();
        const int nToExpand = sh[QT_T + 2];
        for (int i = tid; i < ncand; i += QT_T) {
            const uint32_t kp = kpos[i];
            const int pos = kp & QT_POS_MASK, q = kp >> 28;
            const int r = rnk[pos];
            kpos[i] = (uint32_t)((r >= 0 && r <= jstar) ? newpos4[4 * pos + q] : tailpos[pos]);
        }
        __syncthreads();
        { int4* t4 = bndA; bndA = bndB; bndB = t4; int* t1 = cntA; cntA = cntB; cntB = t1; }
        size = Ctot + U;
        if (size > NC) { if (tid == 0) atomicExch(P.status, PGORB_E_OVERFLOW); size = NC; break; }
        if (size >= N || size == prevSize) break;                          // :669 / :734
        if (!sorted_mode && size + 3 * nToExpand > N) sorted_mode = 1;     // :673
    }

    // ---- best response per node, first in candidate order wins (:741-760) ---------------
    for (int p = tid; p < size; p += QT_T) best[p] = 0ull;
    __syncthreads();
    const int wCell = L.wCell, hCell = L.hCell, nCols = L.nCols;
    for (int i = tid; i < ncand; i += QT_T) {
        const uint32_t cv = cand[i];
        const int x = (cv & 0xFFF) - 3, y = ((cv >> 12) & 0xFFF) - 3;
        const int cj = x / wCell, ci = y / hCell;
        const uint32_t rank = (uint32_t)(((ci * nCols + cj) * hCell + (y - ci * hCell)) * wCell + (x - cj * wCell));
        const unsigned long long key = ((unsigned long long)(cv >> 24) << 32) | (0xFFFFFFFFu - rank);
        atomicMax(&best[kpos[i] & QT_POS_MASK], key);
    }
    __syncthreads();
    uint32_t* sel = P.sel + (int64_t)frame * P.selFrame + L.selOff;
    const int nsel = min(size, L.selCap);
    if (size > L.selCap && tid == 0) atomicExch(P.status, PGORB_E_OVERFLOW);
    for (int i = tid; i < ncand; i += QT_T) {
        const uint32_t cv = cand[i];
        const int x = (cv & 0xFFF) - 3, y = ((cv >> 12) & 0xFFF) - 3;
        const int cj = x / wCell, ci = y / hCell;
        const uint32_t rank = (uint32_t)(((ci * nCols + cj) * hCell + (y - ci * hCell)) * wCell + (x - cj * wCell));
        const unsigned long long key = ((unsigned long long)(cv >> 24) << 32) | (0xFFFFFFFFu - rank);
        const int pos = kpos[i] & QT_POS_MASK;
        if (pos < nsel && best[pos] == key) sel[pos] = cv;
    }
    if (tid == 0) *kpc = nsel;
}

void pg_launch_quadtree(const PgPlan& P, int nframes, hipStream_t s)
{
    int ncMax = 0;
    for (int l = 0; l < P.nlevels; l++) ncMax = max(ncMax, P.lvl[l].nodeCap);
    dim3 grid(P.nlevels, nframes), block(QT_T);
    hipLaunchKernelGGL(k_quadtree, grid, block, (size_t)ncMax * 6 * sizeof(int), s, P);
}

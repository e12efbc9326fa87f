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
//   * the generations need only COUNTS: which children of a node are non-empty / expandable.
//     DivideNode's split points depend on the node's bounds alone, so every key's path through
//     the first D = 5 generations (its depth-D descendant of its root) is pure geometry.  The
//     prologue computes that index once per key and histograms it; summing groups of four gives
//     the counts of every shallower descendant (the "count pyramid", <= 16 KiB of LDS).  While a
//     generation splits depth <= D-1 nodes it runs on the node list and the pyramid alone -- no
//     pass over the ~2e4 keys, and in breadth-first order ONE fused block scan places all
//     children and all survivors.  Deeper trees (corners packed into a few depth-5 cells) leave
//     pyramid mode: a node map turns every key's descendant index into its list position, and
//     each further generation is one pass over the keys that moves them to their new node and
//     counts them into its quadrants (LDS atomics).
//   * at the end a (descendant -> list position) table gives every key its node with one LDS
//     read; best response per node by 64-bit LDS atomicMax on (response, ~candidate order).
//
// The prologue (round 4) is the ONLY pass over the ~2e4 candidates of a tree that ends inside the count pyramid: it reads K2's
// per-cell slots in place (16 lanes per cell: no compaction, no search), histograms every candidate's depth-D descendant and keeps
// the best candidate per descendant; the winner of a final node is then the best of its descendants' winners.  Dense 8-byte key
// records are only built (qt_build_keys: scan over the cells + one binary search per record) when a tree leaves the pyramid.
//
// Integer/compare work on ~2e4 keys; latency-bound (the level-0 workgroup's critical path is
// the kernel), not bandwidth-bound.  tools/experiments/qt_timing.py prints its phases.
#include "pgorb_internal.h"

// Threads per (frame, level) workgroup: chosen per LAUNCH (pg_launch_quadtree_levels: 1 024 when every problem of the launch is resident
// at once -- then the longest problem's latency is the kernel and it wants every wave it can get --, 512 when the problems queue for
// the chip's slots: barriers among 8 waves are cheaper than among 16 and small problems fit four to a CU).  The kernel reads it from blockDim.
#define QT_TMAX 1024
#define QT_T ((int)blockDim.x)
#define QT_SH (__builtin_ctz(blockDim.x))
#define QT_W (QT_T >> 6)
#define QT_POS_MASK 0x0FFFFFFFu

// Wait ONCE for a batch of loads: the values become outputs of an (empty) asm statement, so the
// compiler's waitcnt insertion stops tracking them.  Without this every later use that follows a
// loop and a global store got a full `s_waitcnt vmcnt(0)` -- a store round trip per record.
#ifndef QT_HU
#define QT_HU 4              // records per thread in flight in the prologue (binary searches and slot loads interleaved; 8 measured 2 % slower)
#endif
#define QT_SETTLE4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

// inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (six VALU moves; the __shfl_up form goes through the
// LDS crossbar six times, and the generations pay every one of those in full: they are barrier-to-barrier latency)
__device__ __forceinline__ int qt_wave_incl_scan(int x)
{
    int v = x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31
    return v;
}

// Exclusive prefix sum of a[0..n) in place (a in LDS or global); returns the total.
// All threads must call.  sh: >= QT_W + 1 ints of LDS.
__device__ int qt_scan_excl(int* a, int n, int* sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n + QT_T - 1) >> QT_SH;
    const int b = tid * per, e = min(b + per, n);
    int sum = 0;
    for (int i = b; i < e; i++) sum += a[i];
    const int incl = qt_wave_incl_scan(sum);
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        const int v = (lane < QT_W) ? sh[lane] : 0;
        const int w = qt_wave_incl_scan(v);
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

// Exclusive prefix over the threads of the block of three per-thread values at once (wave
// shuffles + one cross-wave step, 2 barriers).  On return a/b/c hold the exclusive prefixes,
// tot[0..2] the block totals.  sh: >= 3*QT_W + 3 ints.
__device__ __forceinline__ void qt_scan3_threads(int& a, int& b, int& c, int* tot, int* sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ia = qt_wave_incl_scan(a), ib = qt_wave_incl_scan(b), ic = qt_wave_incl_scan(c);
    if (lane == 63) { sh[wv] = ia; sh[QT_W + wv] = ib; sh[2 * QT_W + wv] = ic; }
    __syncthreads();
    int pa = 0, pb = 0, pc = 0, ta = 0, tb = 0, tc = 0;
    // (run-time trip count: the thread count is the launch's.  Unrolled forms -- 16-fold predicated, or 8 / 16 by thread count -- measured 4 .. 30 %
    //  slower on the whole kernel: 48 values in flight do not fit beside the kernel's state under its 64-VGPR cap)
    for (int w = 0; w < QT_W; w++) {
        const int va = sh[w], vb = sh[QT_W + w], vc = sh[2 * QT_W + w];
        if (w < wv) { pa += va; pb += vb; pc += vc; }
        ta += va; tb += vb; tc += vc;
    }
    a = pa + ia - a; b = pb + ib - b; c = pc + ic - c;
    tot[0] = ta; tot[1] = tb; tot[2] = tc;
    __syncthreads();
}

// ctr[target] += 1 for every active lane.  Keys arrive in cell order: when all active lanes of
// the wave hit ONE counter (shallow generations: a handful of counters for ~2e4 keys, plain LDS
// atomics on one address serialise lane by lane) a single lane adds the population count;
// otherwise plain atomics (a per-target leader loop cost ~30 cycles per distinct target and was
// 100 us of the level-0 workgroup).  Must be called by all lanes of the wave.
__device__ __forceinline__ void qt_wave_count(int* ctr, int target, bool active)
{
    const unsigned long long todo = __ballot(active);
    if (todo == 0) return;
    const int leader = __ffsll((long long)todo) - 1;
    const int t = __builtin_amdgcn_readlane(target, leader);
    if (__ballot(active && target != t) == 0) {
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&ctr[t], __popcll(todo));
    } else if (active) {
        atomicAdd(&ctr[target], 1);
    }
}

__device__ __forceinline__ int qt_quadrant(const int4 b, uint32_t cv)
{
    const int midX = b.x + ((b.z - b.x + 1) >> 1);     // UL.x + ceil((UR.x-UL.x)/2)  (:483)
    const int midY = b.y + ((b.w - b.y + 1) >> 1);
    const int x = cv & 0xFFF, y = (cv >> 12) & 0xFFF;
    return (x < midX) ? ((y < midY) ? 0 : 2) : ((y < midY) ? 1 : 3);      // n1 n3 / n2 n4 (:515-526)
}


// Depth-D descendant of the root that holds candidate cv: index r * 4^D + path, where path is the
// sequence of DivideNode quadrants (:481-537) a key at (x, y) falls through.  Pure geometry --
// which descendants exist as list nodes is decided from the counts alone.
__device__ __forceinline__ int qt_leaf(uint32_t cv, float hX, int nIni, int regionH, int D)
{
    const int x = cv & 0xFFF, y = (cv >> 12) & 0xFFF;
    int r = (int)__fdiv_rn((float)x, hX);                // vpIniNodes[kp.pt.x/hX]  (:569)
    r = min(max(r, 0), nIni - 1);
    int x0 = (int)__fmul_rn(hX, (float)r), x1 = (int)__fmul_rn(hX, (float)(r + 1)), y0 = 0, y1 = regionH;
    int idx = r;
    for (int d = 0; d < D; d++) {
        const int midX = x0 + ((x1 - x0 + 1) >> 1), midY = y0 + ((y1 - y0 + 1) >> 1);
        const bool qx = x >= midX, qy = y >= midY;
        idx = idx * 4 + (qx ? 1 : 0) + (qy ? 2 : 0);
        x0 = qx ? midX : x0; x1 = qx ? x1 : midX;
        y0 = qy ? midY : y0; y1 = qy ? y1 : midY;
    }
    return idx;
}

// list position of the node holding the key whose depth-D descendant index is `leaf`
// (map: pyramid-indexed, -1 where no list node sits)
__device__ __forceinline__ int qt_walk(const int* map, int leaf, int nIni, int D)
{
    for (int d = 0; d <= D; d++) {
        const int m = map[qt_pyr_off(nIni, d) + (leaf >> (2 * (D - d)))];
        if (m >= 0) return m;
    }
    return 0;                                            // unreachable: list nodes partition the keys
}

// Dense 8-byte key records {candidate, depth-D descendant} of a (frame, level) from K2's per-cell slots -- what the key
// passes of a tree that grows below the count pyramid work on (rounds 1-3 built them for every problem, in the prologue).
// One thread per OUTPUT record: record i belongs to the last cell whose offset is <= i, a binary search over the scanned
// cell counts in LDS whose first steps are wave-uniform; QT_HU searches per thread run interleaved.  cellOff: cells + 1 ints.
// All threads must call; returns the record count.
// (plain values only: a reference to the kernel's PgPlan argument would put the whole 4-KB struct into scratch memory)
__device__ __forceinline__ int qt_build_keys(const int32_t* __restrict__ cc, const uint32_t* __restrict__ slots, int ncells, int cellCap,
                                             int* cellOff, int* sh, uint2* keys, float hX, int nIni, int regionH, int D)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < ncells; i += QT_T) cellOff[i] = min(cc[i], cellCap);
    __syncthreads();
    const int ncand = qt_scan_excl(cellOff, ncells, sh);
    if (tid == 0) cellOff[ncells] = ncand;
    __syncthreads();
    int steps = 0;
    while ((1 << steps) < ncells) steps++;
    for (int b0 = 0; b0 < ncand; b0 += QT_HU * QT_T) {
        int lo[QT_HU], hi[QT_HU];
#pragma unroll
        for (int u = 0; u < QT_HU; u++) { lo[u] = 0; hi[u] = ncells - 1; }
        for (int st = 0; st < steps; st++) {
#pragma unroll
            for (int u = 0; u < QT_HU; u++) {
                const int mid = (lo[u] + hi[u] + 1) >> 1;
                const bool le = cellOff[mid] <= b0 + u * QT_T + tid;
                lo[u] = le ? mid : lo[u];
                hi[u] = le ? hi[u] : mid - 1;
            }
        }
        uint32_t v[QT_HU];
#pragma unroll
        for (int u = 0; u < QT_HU; u++) {
            const int i = b0 + u * QT_T + tid;
            v[u] = (i < ncand) ? slots[(int64_t)lo[u] * cellCap + (i - cellOff[lo[u]])] : 0u;
        }
        QT_SETTLE4(v[0], v[1], v[2], v[3]);
        if (QT_HU == 8) QT_SETTLE4(v[4 % QT_HU], v[5 % QT_HU], v[6 % QT_HU], v[7 % QT_HU]);
#pragma unroll
        for (int u = 0; u < QT_HU; u++) {
            const int i = b0 + u * QT_T + tid;
            if (i < ncand) keys[i] = make_uint2(v[u], (uint32_t)qt_leaf(v[u], hX, nIni, regionH, D));
        }
    }
    __syncthreads();
    return ncand;
}

// BIG: the level's node arrays do not fit LDS (quota above ~1180) and live in a global slab;
// the same code, just slower.  Each instantiation skips the levels of the other kind.
// WIDE: instantiations for launches that never have more than four waves per SIMD resident -- 512 threads with an LDS need that holds them to
// two workgroups per CU (WIDE 1), 1 024 threads with one workgroup per CU, by LDS or because the launch has no more problems than CUs (WIDE 2):
// 128 VGPRs instead of 64 -- the kernel wants 100 and spills 21 dwords under the cap.
template <bool BIG, int WIDE>
__global__ __launch_bounds__((WIDE == 1 || WIDE == 3) ? 512 : QT_TMAX, WIDE == 3 ? 6 : (WIDE ? 4 : 8)) void k_quadtree(const PgPlan P, int level0, int leafOffInts, int split)
{
    __shared__ int sh[3 * (QT_TMAX / 64) + 8];
    __shared__ int pyr[QT_PYR_CAP];                       // count pyramid, later the node map
    extern __shared__ __attribute__((aligned(16))) int qt_lds[];     // max(30 ints per node, cells + 1)
    const int tid = threadIdx.x;
    // grid = (frames, levels): the heavy level-0 problems of all frames are dispatched first and
    // spread over all CUs (with level as the fast index every 8th workgroup -- always the same
    // 32 CUs under round-robin dispatch -- got all the level-0 work)
    const int l = level0 + blockIdx.y, frame = blockIdx.x;        // (level0: the launch covers levels [level0, level0 + gridDim.y))
    const PgLevel& L = P.lvl[l];
    int* kpc = &P.kpCount[frame * PG_MAXL + l];
    uint2* keys = reinterpret_cast<uint2*>(P.cand) + ((int64_t)frame * P.candFrame + L.candOff);
    if (BIG != (L.nodeOff >= 0)) return;
    const int NC = L.nodeCap;
    int* nodeBase = BIG ? P.nodeScratch + (int64_t)frame * P.nodeFrame + L.nodeOff : qt_lds;
    int4* bndA = reinterpret_cast<int4*>(nodeBase);    // (ULx, ULy, URx, BRy) in list order
    int4* bndB = bndA + NC;
    int* cntA = reinterpret_cast<int*>(bndB + NC);
    int* cntB = cntA + NC;
    int* pidA = cntB + NC;                              // depth << 28 | index at that depth
    int* pidB = pidA + NC;
    int* cnt4 = pidB + NC;                              // [4*NC] keys per child of node p
    int* cnt4n = cnt4 + 4 * NC;                         // [4*NC] the same for the NEXT generation
    int* newpos4 = cnt4n + 4 * NC;                      // [4*NC] new list position of child
    int* tailpos = newpos4 + 4 * NC;                    // [NC]  new position of unprocessed node
    int* rnk = tailpos + NC;                            // [NC]  processing rank or -1
    int* ord = rnk + NC;                                // [NC]  node at processing rank r
    int* cinc = ord + NC;                               // [NC]  inclusive sum of child counts
    // (28 ints per node.)  The two scratch arrays of a "largest first" generation's ranking share newpos4: that is written
    // by the generation's list-writing phase, behind the ranking and a barrier, and read by the key pass that ends the generation
    int* tmp = newpos4;                                 // [NC]
    int* ecnt = newpos4 + NC;                           // [NC]  sizes of expandable nodes
    unsigned long long* best = reinterpret_cast<unsigned long long*>(newpos4);    // [NC] (epilogue)
    // aux area behind the node arrays (max(leaves, cells + 1) ints): the best candidate of every depth-D descendant,
    // (response << 24 | ~order rank), while the tree is inside the count pyramid; the scanned cell counts while key records are built
    uint32_t* leafBest = reinterpret_cast<uint32_t*>(qt_lds + leafOffInts);
    int* cellOff = qt_lds + leafOffInts;

    const int N = L.quota;
    const int regionH = L.h - 2 * PG_EDGE;              // maxY - minY
    const int nIni = L.nIni;
    const float hX = L.hX;
    const int D = qt_pyr_depth(nIni);
    const int pyrTotal = qt_pyr_off(nIni, D + 1);

    // candidate order = (cell row, cell col, y, x); x / wCell by an exact reciprocal (x < 4096, wCell < 256)
    const int wCell = L.wCell, hCell = L.hCell, nCols = L.nCols;
    const uint32_t mW = ((1u << 20) + wCell - 1) / wCell, mH = ((1u << 20) + hCell - 1) / hCell;
    auto order_rank = [&](uint32_t cv) {
        const uint32_t x = (cv & 0xFFF) - 3, y = ((cv >> 12) & 0xFFF) - 3;
        const uint32_t cj = (x * mW) >> 20, ci = (y * mH) >> 20;
        return ((ci * nCols + cj) * hCell + (y - ci * hCell)) * wCell + (x - cj * wCell);
    };
    // the 64-bit key a candidate bids for its node with: (response, first in candidate order wins)
    auto bid = [&](uint32_t cv) { return ((unsigned long long)(cv >> 24) << 32) | (0xFFFFFFFFu - order_rank(cv)); };
    // ... and its 32-bit form, when every order rank of the level fits 24 bits (levels up to ~16 Mpx: all the reference runs)
    const bool rank24 = (long long)L.nCols * L.nRows * hCell * wCell <= (1 << 24);

    // ---- prologue (round 4): K2's per-cell slots -> leaf histogram + best candidate per leaf, NO key records ------------
    // A candidate's path through the first D generations is pure geometry, the generations need only counts, and the winner of
    // a final node is the best of the winners of its depth-D descendants -- so a tree that ends inside the count pyramid (the
    // common case) never needs its ~2e4 candidates again: this pass is the only one over them.  (Rounds 1-3 compacted the slots
    // into dense 8-byte key records first -- a scan over the cells, one binary search per record, 31 MB written and read twice
    // per step -- and ran a second pass over all records for the winners: 71 of the level-0 workgroup's 107 us.)  Trees that
    // grow deeper build the key records when they leave the pyramid (qt_build_keys).
    // Mapping: 16 lanes per cell, 4 cells per wave and round -- one 64-byte sector of a cell's slots per 16 lanes, no search, no
    // scan; cells with more than 16 records take more rounds.  Two groups of cells are in flight per wave.
    int ncand;
    bool keysBuilt = false;
    {
        const int ncells = L.nCols * L.nRows;
        const int32_t* cc = P.cellCount + (int64_t)frame * P.totalCells + L.cellBase;
        const uint32_t* slots = P.cellCand + (int64_t)frame * P.cellCandFrame + L.cellCandOff;
        const int nleaf = nIni << (2 * D);
        for (int i = tid; i < pyrTotal; i += QT_T) pyr[i] = 0;
        if (!split) for (int i = tid; i < nleaf; i += QT_T) leafBest[i] = 0u;
        if (tid == 0) sh[QT_W + 2] = 0;
        int mine = 0;                                       // records counted by this lane
        const int lane = tid & 63;
        if (split) {
            // two launches: k_qt_leaves (below) has run the pass on many small workgroups; take the leaves' counts and best records
            __syncthreads();
            int* hDs = pyr + qt_pyr_off(nIni, D);
            uint2* gl = P.qtLeaf + ((int64_t)frame * P.nlevels + l) * PG_QT_LEAF_CAP;
            for (int i = tid; i < nleaf; i += QT_T) {
                const uint2 e = gl[i];
                hDs[i] = (int)e.x; leafBest[i] = e.y; mine += (int)e.x;
            }
        } else {
        // Both things the pass needs of a candidate are SEPARABLE in x and y: DivideNode splits a node at the middle of its x range
        // and of its y range independently (:483-485), so the depth-D descendant index is (root and x-path bits) | (y-path bits);
        // and the candidate-order rank ((cell row * nCols + cell col) * hCell + row in cell) * wCell + col in cell is a y term
        // plus an x term.  Two tables of (descendant part, rank part) per region column / row, built here by the workgroup
        // (3 entries per thread at 1080p) in the LDS the node arrays will take later, turn ~65 instructions per candidate
        // (five split steps, two reciprocal divisions) into two 8-byte LDS reads, an OR and an add.
        const int regionW = L.w - 2 * PG_EDGE;
        uint2* xTab = reinterpret_cast<uint2*>(qt_lds);           // [regionW]
        uint2* yTab = xTab + regionW;                             // [regionH]
        for (int i = tid; i < regionW + regionH; i += QT_T) {
            const bool isY = i >= regionW;
            const int c = isY ? i - regionW : i;                  // the coordinate as K2 stores it (region-relative)
            (isY ? yTab : xTab)[c] = qt_tab_entry(isY, c, hX, nIni, regionH, D, wCell, hCell, nCols);
        }
        __syncthreads();
        int* hD = pyr + qt_pyr_off(nIni, D);
        // (the wave index as a scalar: with cb / cEnd in VGPRs the compiler made the cell loop a per-lane loop around the cross-lane
        //  shuffles below, and that build hung the kernel -- found by a timed-out run, round 4)
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int cellCap = L.cellCap;
        // One LANE per cell, 64 cells per wave step, steps dealt round-robin to the 16 waves; ALL of a cell's records (up to 32 per
        // round; a cell rarely holds more) are in flight before the first is used, 16 bytes per load.  The first form of this pass
        // (16 lanes per cell, 32 cells per step, one round of 16 slots per trip) spent its 27 us per level-0 problem waiting: ten
        // dependent memory round trips per wave; this one makes three.  The 64 lanes of an LDS atomic now belong to 64 different
        // cells -- mostly different leaves.  (A cell's slots are read up to 3 records past its count: inside the slab, api.hip.)
        struct __attribute__((packed, aligned(4))) U4 { uint32_t e[4]; };
        if (ncells < QT_T) {
            // fewer cells than threads (the upper levels, small frames): 16 lanes per cell, 32 cells per wave step keeps the workgroup's
            // lanes busy where one lane per cell would leave most of them idle (640 x 480, level 0: 300 cells) -- the round-4 form
            const int sub = lane >> 4, sl = lane & 15;
            const int cpw = (((ncells + QT_W - 1) >> (QT_SH - 6)) + 31) & ~31;
            const int cBeg = wv * cpw, cEnd = min(ncells, cBeg + cpw);
            int cntNext = (lane < 32 && cBeg + lane < cEnd) ? cc[cBeg + lane] : 0;
            for (int cb = cBeg; cb < cEnd; cb += 32) {
                const int myc = min(cntNext, cellCap);
                cntNext = (lane < 32 && cb + 32 + lane < cEnd) ? cc[cb + 32 + lane] : 0;
                mine += myc;
                int n[8];
                int maxn = myc;
#pragma unroll
                for (int g = 0; g < 8; g++) n[g] = __shfl(myc, 4 * g + sub);
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) maxn = max(maxn, __shfl_xor(maxn, d));
                maxn = __builtin_amdgcn_readfirstlane(maxn);
                const uint32_t* sp = slots + (int64_t)(cb + sub) * cellCap + sl;
                for (int base = 0; base < maxn; base += 16) {       // (wave-uniform trip count)
                    uint32_t v[8];
#pragma unroll
                    for (int g = 0; g < 8; g++) v[g] = (base + sl < n[g]) ? sp[(int64_t)(4 * g) * cellCap + base] : 0u;
                    QT_SETTLE4(v[0], v[1], v[2], v[3]); QT_SETTLE4(v[4], v[5], v[6], v[7]);
#pragma unroll
                    for (int g = 0; g < 8; g++) {
                        if (base + sl < n[g]) {
                            const uint2 tx = xTab[v[g] & 0xFFF], ty = yTab[(v[g] >> 12) & 0xFFF];
                            const int lf = (int)(tx.x | ty.x);
                            atomicAdd(&hD[lf], 1);
                            if (rank24) atomicMax(&leafBest[lf], (v[g] & 0xFF000000u) | (0xFFFFFFu - (tx.y + ty.y)));
                        }
                    }
                }
            }
        } else {
        const int cFirst = 64 * wv;
        int cntNext = (cFirst + lane < ncells) ? cc[cFirst + lane] : 0;
        for (int cb = cFirst; cb < ncells; cb += 64 * QT_W) {
            const int myc = min(cntNext, cellCap);
            cntNext = (cb + 64 * QT_W + lane < ncells) ? cc[cb + 64 * QT_W + lane] : 0;
            mine += myc;
            int maxn = myc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) maxn = max(maxn, __shfl_xor(maxn, d));
            maxn = __builtin_amdgcn_readfirstlane(maxn);
            const uint32_t* sp = slots + (int64_t)(cb + lane) * cellCap;      // (lanes past the last cell: myc = 0, never dereferenced)
            for (int base = 0; base < maxn; base += 32) {         // (wave-uniform trip count)
                U4 r[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    r[j] = U4{{0u, 0u, 0u, 0u}};
                    if (base + 4 * j < myc) r[j] = *reinterpret_cast<const U4*>(sp + base + 4 * j);
                }
                const int lim = min(maxn - base, 32);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (4 * j < lim) {                            // (wave-uniform)
                        uint2 tx[4], ty[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool ok = base + 4 * j + k < myc;
                            const uint32_t v = ok ? r[j].e[k] : 0u;  // (a slot past the count holds anything: entry 0 of the tables instead)
                            tx[k] = xTab[v & 0xFFF]; ty[k] = yTab[(v >> 12) & 0xFFF];
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (base + 4 * j + k < myc) {
                                const int lf = (int)(tx[k].x | ty[k].x);
                                atomicAdd(&hD[lf], 1);
                                if (rank24) atomicMax(&leafBest[lf], (r[j].e[k] & 0xFF000000u) | (0xFFFFFFu - (tx[k].y + ty[k].y)));
                            }
                        }
                    }
                }
            }
        }
        }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
        if (lane == 0 && mine) atomicAdd(&sh[QT_W + 2], mine);
        __syncthreads();
        ncand = sh[QT_W + 2];
        if (tid == 0) P.candCount[frame * PG_MAXL + l] = ncand;
        if (ncand <= 0 || nIni < 1) {
            // no keypoint on this level: every slot of its selection slab is marked unused for K4-6
            PgSelRec* selE = reinterpret_cast<PgSelRec*>(P.sel) + ((int64_t)frame * P.selFrame + L.selOff);
            for (int p = tid; p < L.selCap; p += QT_T) selE[p].posLevel = 0xFFFFFFFFu;
            if (tid == 0) *kpc = 0;
            if (ncand > 0 && tid == 0) atomicExch(P.status, PGORB_E_TOOSMALL);      // see api.hip level_geometry: reference UB, reported
            return;
        }
        if (!rank24) {                                      // order ranks beyond 24 bits: the winners come from a key pass (64-bit bids)
            qt_build_keys(cc, slots, ncells, L.cellCap, cellOff, sh, keys, hX, nIni, regionH, D);
            keysBuilt = true;
        }
        // counts of the shallower descendants
        for (int d = D - 1; d >= 0; d--) {
            const int* src = pyr + qt_pyr_off(nIni, d + 1);
            int* dst = pyr + qt_pyr_off(nIni, d);
            const int cnt = nIni << (2 * d);
            for (int j = tid; j < cnt; j += QT_T) dst[j] = src[4 * j] + src[4 * j + 1] + src[4 * j + 2] + src[4 * j + 3];
            __syncthreads();
        }
    }

    // ---- initial nodes (:543-585) ------------------------------------------------------
    int size;
    {
        if (tid == 0) {
            int pos = 0;
            for (int i = 0; i < nIni; i++) {
                if (pyr[i] > 0) {
                    bndA[pos] = make_int4((int)__fmul_rn(hX, (float)i), 0,
                                          (int)__fmul_rn(hX, (float)(i + 1)), regionH);
                    cntA[pos] = pyr[i];
                    pidA[pos] = i;
                    pos++;
                }
            }
            sh[QT_W + 1] = pos;
        }
        __syncthreads();
        size = sh[QT_W + 1];
        if (D >= 1)
            for (int i = tid; i < 4 * size; i += QT_T) cnt4[i] = pyr[nIni + 4 * pidA[i >> 2] + (i & 3)];
        __syncthreads();
    }

    // ---- generations (:594-739) ---------------------------------------------------------
    // Generation g splits depth g-1 nodes.  While g <= D the child counts come from the pyramid
    // and no pass over the keys is needed ("pyramid mode"); deeper trees continue with one key
    // pass per generation (keys carry node position | quadrant << 28).
    int sorted_mode = 0, gen = 1;
    bool last = false, pyrMode = true;
    while (!last) {
        if (pyrMode && gen > D) {
            // leave pyramid mode: the dense key records (only now: the pyramid generations did not need them), the node map,
            // then every key finds its node and is counted into its quadrant
            if (!keysBuilt) {
                qt_build_keys(P.cellCount + (int64_t)frame * P.totalCells + L.cellBase, P.cellCand + (int64_t)frame * P.cellCandFrame + L.cellCandOff,
                              L.nCols * L.nRows, L.cellCap, cellOff, sh, keys, hX, nIni, regionH, D);       // (cellOff shares the aux area with leafBest, which is dead from here on)
                keysBuilt = true;
            }
            for (int i = tid; i < pyrTotal; i += QT_T) pyr[i] = -1;
            for (int i = tid; i < 4 * size; i += QT_T) cnt4[i] = 0;
            __syncthreads();
            for (int p = tid; p < size; p += QT_T) {
                const int pid = pidA[p];
                pyr[qt_pyr_off(nIni, pid >> 28) + (pid & QT_POS_MASK)] = p;
            }
            __syncthreads();
            for (int i0 = 0; i0 < ncand; i0 += QT_T) {
                const int i = i0 + tid;
                bool act = i < ncand;
                int tgt = 0;
                if (act) {
                    const uint2 k = keys[i];
                    const int pos = qt_walk(pyr, (int)k.y, nIni, D);
                    uint32_t rec = (uint32_t)pos;
                    act = cntA[pos] > 1;
                    if (act) {
                        const int q = qt_quadrant(bndA[pos], k.x);
                        tgt = pos * 4 + q;
                        rec |= (uint32_t)q << 28;
                    }
                    keys[i].y = rec;
                }
                qt_wave_count(cnt4, tgt, act);
            }
            __syncthreads();
            pyrMode = false;
        }
        const int n = size, prevSize = size;
        const bool fill = pyrMode && gen + 1 <= D;       // children get their counts from the pyramid
        const int fillOff = qt_pyr_off(nIni, min(gen + 1, D));
        int jstar = 0;
        if (!sorted_mode) {
            // ---- breadth-first generation (:606-665): processing order == list order and every
            // expandable node is split, so ONE fused scan places all children and all survivors.
            const int per = (n + QT_T - 1) >> QT_SH;
            const int pb = tid * per, pe = min(pb + per, n);
            int se = 0, sc = 0, sx = 0;
            for (int p = pb; p < pe; p++)
                if (cntA[p] > 1) {
                    const int c0 = cnt4[4 * p], c1 = cnt4[4 * p + 1], c2 = cnt4[4 * p + 2], c3 = cnt4[4 * p + 3];
                    se++; sc += (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0); sx += (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                }
            int tot[3];
            qt_scan3_threads(se, sc, sx, tot, sh);           // se, sc = exclusive prefixes of this thread's chunk
            const int m = tot[0], Ctot = tot[1], nToExpand = tot[2];
            size = Ctot + (n - m);
            if (size > NC) { if (tid == 0) atomicExch(P.status, PGORB_E_OVERFLOW); size = min(size, NC); last = true; }
            if (size >= N || size == prevSize) last = true;                        // :669
            else if (size + 3 * nToExpand > N) sorted_mode = 1;                    // :673
            int re = se, ce = sc;
            for (int p = pb; p < pe; p++) {
                if (cntA[p] > 1) {
                    const int4 b = bndA[p];
                    const int midX = b.x + ((b.z - b.x + 1) >> 1);
                    const int midY = b.y + ((b.w - b.y + 1) >> 1);
                    const int cq[4] = {cnt4[4 * p], cnt4[4 * p + 1], cnt4[4 * p + 2], cnt4[4 * p + 3]};
                    const int c = (cq[0] > 0) + (cq[1] > 0) + (cq[2] > 0) + (cq[3] > 0);
                    const int cidx = (int)(((uint32_t)pidA[p] & (QT_POS_MASK >> 2)) * 4u);
                    int pos = Ctot - (ce + c);               // children of later parents are in front (:623-660)
                    if (pos >= 0 && pos + c <= NC) {
#pragma unroll
                        for (int q = 3; q >= 0; q--) {       // push_front order n1..n4 => n4 first
                            if (cq[q] <= 0) continue;
                            bndB[pos] = make_int4((q & 1) ? midX : b.x, (q & 2) ? midY : b.y,
                                                  (q & 1) ? b.z : midX, (q & 2) ? b.w : midY);
                            cntB[pos] = cq[q];
                            pidB[pos] = (int)(((uint32_t)gen << 28) | (uint32_t)(cidx + q));
                            if (fill && cq[q] > 1) {
                                const int* h = pyr + fillOff + (cidx + q) * 4;
                                cnt4n[4 * pos] = h[0]; cnt4n[4 * pos + 1] = h[1]; cnt4n[4 * pos + 2] = h[2]; cnt4n[4 * pos + 3] = h[3];
                            }
                            newpos4[4 * p + q] = pos++;
                        }
                    }
                    rnk[p] = 0; re++; ce += c;
                } else {
                    const int pos = min(Ctot + (p - re), NC - 1);   // single-key nodes keep their order behind
                    bndB[pos] = bndA[p]; cntB[pos] = cntA[p]; pidB[pos] = pidA[p];
                    tailpos[p] = pos; rnk[p] = -1;
                }
            }
        } else {
            // ---- "largest first" generation (:676-737) ----
            for (int p = tid; p < n; p += QT_T) rnk[p] = (cntA[p] > 1) ? 1 : 0;
            __syncthreads();
            const int m = qt_scan_excl(rnk, n, sh);          // rnk[p] = list-order rank among expandable
            if (BIG) {
                // descending size, ties: smaller list position first (see header); node counts beyond
                // the 11-bit rank field of the packed key below
                for (int p = tid; p < n; p += QT_T) {
                    if (cntA[p] > 1) { tmp[rnk[p]] = p; ecnt[rnk[p]] = cntA[p]; }
                    else rnk[p] = -1;
                }
                __syncthreads();
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
                __syncthreads();
            } else {
                // sort key of an expandable node: size << 11 | (2047 - list-order rank), so that "larger
                // size first, ties: smaller list position first" (see header) is one unsigned compare
                for (int p = tid; p < n; p += QT_T) {
                    if (cntA[p] > 1) { tmp[rnk[p]] = p; ecnt[rnk[p]] = (int)(((uint32_t)min(cntA[p], 0x1FFFFF) << 11) | (uint32_t)(2047 - rnk[p])); }
                    else rnk[p] = -1;
                }
                __syncthreads();
                // rank by counting: m <= nodeCap < 2048 keys, four per LDS read
                const int m4 = m & ~3;
                for (int i = tid; i < m; i += QT_T) {
                    const uint32_t ki = (uint32_t)ecnt[i];
                    int rank = 0;
                    for (int j = 0; j < m4; j += 4) {
                        const uint4 k4 = *reinterpret_cast<const uint4*>(ecnt + j);
                        rank += (k4.x > ki) + (k4.y > ki) + (k4.z > ki) + (k4.w > ki);
                    }
                    for (int j = m4; j < m; j++) rank += (uint32_t)ecnt[j] > ki;
                    ord[rank] = tmp[i];
                    rnk[tmp[i]] = rank;
                }
                __syncthreads();
            }
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
            if (tid == 0) sh[QT_W + 1] = m - 1;
            __syncthreads();
            for (int r = tid; r < m; r += QT_T) {            // early break at N (:730)
                const bool now = n + cinc[r] - (r + 1) >= N;
                const bool before = (r > 0) && (n + cinc[r - 1] - r >= N);
                if (now && !before) sh[QT_W + 1] = r;
            }
            __syncthreads();
            jstar = sh[QT_W + 1];
            const int Ctot = (m > 0) ? cinc[jstar] : 0;
            // unprocessed nodes keep their relative order behind the new children
            for (int p = tid; p < n; p += QT_T) tailpos[p] = (rnk[p] >= 0 && rnk[p] <= jstar) ? 0 : 1;
            __syncthreads();
            const int U = qt_scan_excl(tailpos, n, sh);
            // write the new list
            for (int p = tid; p < n; p += QT_T) {
                const int r = rnk[p];
                if (r >= 0 && r <= jstar) {
                    const int4 b = bndA[p];
                    const int midX = b.x + ((b.z - b.x + 1) >> 1);
                    const int midY = b.y + ((b.w - b.y + 1) >> 1);
                    const int cq[4] = {cnt4[4 * p], cnt4[4 * p + 1], cnt4[4 * p + 2], cnt4[4 * p + 3]};
                    const int cidx = (int)(((uint32_t)pidA[p] & (QT_POS_MASK >> 2)) * 4u);
                    int pos = Ctot - cinc[r];                // children of later-processed parents are in front
#pragma unroll
                    for (int q = 3; q >= 0; q--) {
                        if (cq[q] <= 0) continue;
                        bndB[pos] = make_int4((q & 1) ? midX : b.x, (q & 2) ? midY : b.y,
                                              (q & 1) ? b.z : midX, (q & 2) ? b.w : midY);
                        cntB[pos] = cq[q];
                        pidB[pos] = (int)(((uint32_t)gen << 28) | (uint32_t)(cidx + q));
                        if (fill && cq[q] > 1) {
                            const int* h = pyr + fillOff + (cidx + q) * 4;
                            cnt4n[4 * pos] = h[0]; cnt4n[4 * pos + 1] = h[1]; cnt4n[4 * pos + 2] = h[2]; cnt4n[4 * pos + 3] = h[3];
                        }
                        newpos4[4 * p + q] = pos++;
                    }
                } else {
                    const int pos = Ctot + tailpos[p];
                    bndB[pos] = bndA[p]; cntB[pos] = cntA[p]; pidB[pos] = pidA[p];
                    tailpos[p] = pos;
                    if (fill && cntA[p] > 1) {               // not reached in this generation: counts carry over
                        cnt4n[4 * pos] = cnt4[4 * p]; cnt4n[4 * pos + 1] = cnt4[4 * p + 1];
                        cnt4n[4 * pos + 2] = cnt4[4 * p + 2]; cnt4n[4 * pos + 3] = cnt4[4 * p + 3];
                    }
                }
            }
            size = Ctot + U;
            if (size > NC) { if (tid == 0) atomicExch(P.status, PGORB_E_OVERFLOW); size = NC; last = true; }
            if (size >= N || size == prevSize) last = true;                    // :734
        }
        if (!pyrMode) {
            // keys move to their new node; unless this was the last generation they are also
            // counted into that node's quadrants for the next generation
            for (int i = tid; i < 4 * size; i += QT_T) cnt4n[i] = 0;
            __syncthreads();
            // 4 keys per thread and step: the four independent global loads are in flight together
            for (int b0 = 0; b0 < ncand; b0 += 4 * QT_T) {
                uint2 kk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = b0 + tid + u * QT_T; kk[u] = (i < ncand) ? keys[i] : make_uint2(0u, 0u); }
        QT_SETTLE4(kk[0].x, kk[1].x, kk[2].x, kk[3].x); QT_SETTLE4(kk[0].y, kk[1].y, kk[2].y, kk[3].y);
                int tgt[4];
                bool act[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = b0 + tid + u * QT_T;
                    act[u] = i < ncand;
                    tgt[u] = 0;
                    if (act[u]) {
                        const uint2 k = kk[u];
                        const int pos = k.y & QT_POS_MASK, q = k.y >> 28;
                        const int r = rnk[pos];
                        const int np = (r >= 0 && r <= jstar) ? newpos4[4 * pos + q] : tailpos[pos];
                        uint32_t rec = (uint32_t)np;
                        act[u] = !last && cntB[np] > 1;
                        if (act[u]) {
                            const int q2 = qt_quadrant(bndB[np], k.x);
                            tgt[u] = np * 4 + q2;
                            rec |= (uint32_t)q2 << 28;
                        }
                        // the last generation's pass is the last time the keys are looked at: each key bids for its final node
                        // here (the quadrant counters of a generation that will not come are free: 2 ints per node, zeroed above)
                        // instead of being filed for one more pass over all of them
                        if (last) atomicMax(&reinterpret_cast<unsigned long long*>(cnt4n)[np], bid(k.x));
                        else keys[i].y = rec;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) qt_wave_count(cnt4n, tgt[u], act[u]);
            }
        }
        __syncthreads();
        { int4* t4 = bndA; bndA = bndB; bndB = t4; int* t1 = cntA; cntA = cntB; cntB = t1;
          t1 = cnt4; cnt4 = cnt4n; cnt4n = t1; t1 = pidA; pidA = pidB; pidB = t1; }
        gen++;
    }

    // ---- best response per node, first in candidate order wins (:741-760) ---------------
    // A tree that ended below the count pyramid has its bids already (the last generation's key pass; after the swap above they
    // sit in cnt4).  A tree that ended inside the pyramid never looked at its keys again: one pass now, every key to the list
    // position of its depth-D descendant.  (best aliases newpos4 there: the generation loop ended with a barrier)
    if (pyrMode) {
        for (int p = tid; p < size; p += QT_T) best[p] = 0ull;
        int* leafPos = pyr + qt_pyr_off(nIni, D);             // depth-D descendant -> list position
        for (int i = tid; i < pyrTotal; i += QT_T) pyr[i] = -1;
        __syncthreads();
        for (int p = tid; p < size; p += QT_T) {
            const int pid = pidA[p];
            pyr[qt_pyr_off(nIni, pid >> 28) + (pid & QT_POS_MASK)] = p;
        }
        __syncthreads();
        const int nleaf = nIni << (2 * D);
        if (!keysBuilt) {
            // the winner of a node = the best of its leaves' winners (prologue): one step per LEAF (<= 3072), not per candidate
            for (int j = tid; j < nleaf; j += QT_T) {
                const uint32_t k = leafBest[j];
                if (k) atomicMax(&best[qt_walk(pyr, j, nIni, D)], ((unsigned long long)(k >> 24) << 32) | (0xFF000000u | (k & 0xFFFFFFu)));   // the 64-bit bid
            }
        } else {
            for (int j = tid; j < nleaf; j += QT_T) leafPos[j] = qt_walk(pyr, j, nIni, D);
            __syncthreads();
            for (int b0 = 0; b0 < ncand; b0 += 4 * QT_T) {
                uint2 kk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = b0 + tid + u * QT_T; kk[u] = (i < ncand) ? keys[i] : make_uint2(0u, 0u); }
                QT_SETTLE4(kk[0].x, kk[1].x, kk[2].x, kk[3].x); QT_SETTLE4(kk[0].y, kk[1].y, kk[2].y, kk[3].y);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = b0 + tid + u * QT_T;
                    if (i >= ncand) break;
                    atomicMax(&best[leafPos[kk[u].y]], bid(kk[u].x));
                }
            }
        }
    } else {
        best = reinterpret_cast<unsigned long long*>(cnt4);
    }
    __syncthreads();
    // Selection records are written in a DISPATCH order for K4-6 -- 64-px (or coarser) tiles, row-major -- as pairs
    // {record, position in the reference's output list}: K4-6 gives XCD x the x-th eighth of this order, so the
    // 43 x 43 windows one XCD's L2 sees overlap (in list order every window line was fetched from HBM about twice).
    // The keypoint's OUTPUT slot stays its list position.  Counting sort on the tile index, in the pyramid's LDS.
    PgSelRec* sel = reinterpret_cast<PgSelRec*>(P.sel) + ((int64_t)frame * P.selFrame + L.selOff);
    PgSelRec proto;
    proto.cv = 0; proto.posLevel = (uint32_t)l << 16; proto.pitch = L.pitch; proto.wh = (uint32_t)L.w | ((uint32_t)L.h << 16);
    proto.plane = (uint64_t)(uintptr_t)(L.img + (int64_t)frame * L.fstride); proto.scale = L.scale; proto.patchSize = L.patchSize;
    const int nsel = min(size, L.selCap);
    if (size > L.selCap && tid == 0) atomicExch(P.status, PGORB_E_OVERFLOW);
    const int tileCap = min(1024, 4 * NC - 2);              // the histogram lives in cnt4 [4 * NC]
    int tsh = 6;
    while ((((L.w - 2 * PG_EDGE) >> tsh) + 1) * (((L.h - 2 * PG_EDGE) >> tsh) + 1) > tileCap) tsh++;
    const int tilesX = ((L.w - 2 * PG_EDGE) >> tsh) + 1;
    int* hist = pyrMode ? cnt4 : cnt4n;                     // generation state is dead in the epilogue (cnt4 may hold the bids)
    int* wRec = cntA;                                       // [NC] winner record of list position p
    int* wKey = cntB;                                       // [NC] tile << 16 | arrival index inside the tile
    for (int i = tid; i <= tileCap; i += QT_T) hist[i] = 0;
    __syncthreads();
    // The winner of a node is named by best[p] itself: the key holds the response and the candidate-order rank, and the rank is
    // (cell, row in the cell, column in the cell) -- so x, y come back out of it and no third pass over the ~2e4 key records is
    // needed to find the winners (that pass was 18 of the level-0 workgroup's 123 us).
    for (int p = tid; p < nsel; p += QT_T) {
        const unsigned long long b = best[p];
        uint32_t cv = 0u;
        if (b) {                                            // (every node of the final list holds a key)
            const uint32_t rank = 0xFFFFFFFFu - (uint32_t)b;
            const uint32_t t1 = rank / (uint32_t)wCell, xr = rank - t1 * (uint32_t)wCell;
            const uint32_t cidx = t1 / (uint32_t)hCell, yr = t1 - cidx * (uint32_t)hCell;
            const uint32_t ci = cidx / (uint32_t)nCols, cj = cidx - ci * (uint32_t)nCols;
            cv = (cj * wCell + xr + 3u) | ((ci * hCell + yr + 3u) << 12) | ((uint32_t)(b >> 32) << 24);
        }
        const int t = (int)(((cv >> 12) & 0xFFF) >> tsh) * tilesX + (int)((cv & 0xFFF) >> tsh);
        wRec[p] = (int)cv;
        wKey[p] = (t << 16) | atomicAdd(&hist[t], 1);
    }
    __syncthreads();
    qt_scan_excl(hist, tileCap + 1, sh);
    for (int p = tid; p < nsel; p += QT_T) {
        const int wk = wKey[p];
        PgSelRec r = proto;
        r.cv = (uint32_t)wRec[p]; r.posLevel |= (uint32_t)p;
        sel[hist[wk >> 16] + (wk & 0xFFFF)] = r;
    }
    for (int p = nsel + tid; p < L.selCap; p += QT_T) sel[p].posLevel = 0xFFFFFFFFu;       // unused slots of the slab: K4-6 returns on them
    if (tid == 0) *kpc = nsel;
}

// K3's candidate pass as its own launch (round 4, PgPlan::qtSplit): inside k_quadtree it is ONE workgroup's walk over up to 9e4
// candidates (27 us of a 1080p level-0 problem, ~100 us at 2160p, with the workgroup's other phases waiting behind it); here a
// workgroup OWNS a few leaf rows of a (frame, level) problem -- a leaf row = the depth-D descendants that share their y path, a
// band of the region ~34 px high at 1080p -- and walks the cell rows that overlap the band (about QTP_CELLS cells; a cell row that
// straddles two bands is read by both owners, records outside the band are skipped).  Ownership means no atomics to HBM and no
// state between batches: every leaf of the problem is written, zero or not, by exactly one workgroup.  Coordinate tables come from
// the plan ({leaf column | row, rank part} per region column | row) and are copied to LDS: the x table whole, the y table for the band.
#define QTP_T 512
__global__ __launch_bounds__(QTP_T) void k_qt_leaves(const PgPlan P, int level0, int nlev)
{
    extern __shared__ __attribute__((aligned(16))) int qp_lds[];     // [regionW] x table, [band] y table (uint2), [rows * ncol] counts, best records
    const int tid = threadIdx.x, frame = blockIdx.y, lane = tid & 63;
    int rem = blockIdx.x, l = level0;
    for (int k = 0; k < nlev; k++) {
        const PgLevel& V = P.lvl[level0 + k];
        const int g = qt_pass_groups(V.nCols * V.nRows, qt_pyr_depth(V.nIni));
        l = level0 + k;
        if (rem < g) break;
        rem -= g;
    }
    const PgLevel& L = P.lvl[l];
    const int nIni = L.nIni, nCols = L.nCols, ncells = nCols * L.nRows, cellCap = L.cellCap, hCell = L.hCell;
    const int32_t* cc = P.cellCount + (int64_t)frame * P.totalCells + L.cellBase;
    if (nIni < 1) {                                          // no root: the reference's behaviour is undefined there (api.hip level_geometry), reported
        if (rem == 0) {
            int any = 0;
            for (int i = tid; i < ncells; i += QTP_T) any |= cc[i];
            if (any) atomicExch(P.status, PGORB_E_TOOSMALL);
        }
        return;
    }
    const int D = qt_pyr_depth(nIni), rows = 1 << D, ncol = nIni << D;
    const int rowsPer = qt_pass_rows_per_group(ncells, D);
    const int r0 = rem * rowsPer, r1 = min(rows, r0 + rowsPer);
    const int regionW = L.w - 2 * PG_EDGE, regionH = L.h - 2 * PG_EDGE;
    const uint2* gx = P.qtTab + L.qtTabOff;
    const uint2* gy = gx + regionW;
    const uint2* gr = gy + regionH;
    const int ya = (int)gr[r0].x, yb = (int)gr[r1].x;        // the band, in K2's (region-relative) coordinates
    uint2* xT = reinterpret_cast<uint2*>(qp_lds);
    uint2* yT = xT + regionW;                                // [yb - ya], indexed y - ya
    const int nh = (r1 - r0) * ncol;
    int* hC = reinterpret_cast<int*>(yT + (yb - ya));        // [nh] counts + one dump bin per lane (records outside the band land there:
    uint32_t* hB = reinterpret_cast<uint32_t*>(hC + nh + 64); //  the inner loop has no branches, so its LDS reads and atomics pipeline)
    for (int i = tid; i < regionW; i += QTP_T) xT[i] = gx[i];
    for (int i = tid; i < yb - ya; i += QTP_T) { uint2 e = gy[ya + i]; e.x = (e.x - (uint32_t)r0) * (uint32_t)ncol; yT[i] = e; }    // (row offset in the histogram)
    for (int i = tid; i < 2 * (nh + 64); i += QTP_T) hC[i] = 0;
    const uint32_t* slots = P.cellCand + (int64_t)frame * P.cellCandFrame + L.cellCandOff;
    const bool rank24 = (long long)ncells * hCell * L.wCell <= (1 << 24);
    // a record of cell row ci has y in [ci * hCell + 3, (ci + 1) * hCell + 3)  (K2's coordinates start at 3)
    const int ciBeg = max(ya - 3, 0) / hCell, ciEnd = min(L.nRows - 1, max(yb - 1 - 3, 0) / hCell);
    const int c0 = ciBeg * nCols, c1 = (yb > ya) ? (ciEnd + 1) * nCols : c0;
    // One LANE per cell, 64 cells per wave step, steps dealt round-robin to the waves (the last, partly filled step exists once per
    // workgroup, not once per wave): the 64 lanes of an LDS atomic belong to 64 different cells, i.e. mostly different leaves.
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cFirst = c0 + 64 * wv;
    int cntNext = (cFirst + lane < c1) ? cc[cFirst + lane] : 0;
    __syncthreads();
    struct __attribute__((packed, aligned(4))) U4 { uint32_t e[4]; };
    for (int cb = cFirst; cb < c1; cb += 64 * (QTP_T / 64)) {
        const int myc = min(cntNext, cellCap);
        cntNext = (cb + 64 * (QTP_T / 64) + lane < c1) ? cc[cb + 64 * (QTP_T / 64) + lane] : 0;
        int maxn = myc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) maxn = max(maxn, __shfl_xor(maxn, d));
        maxn = __builtin_amdgcn_readfirstlane(maxn);
        const uint32_t* sp = slots + (int64_t)(cb + lane) * cellCap;      // (lanes past c1: myc = 0, never dereferenced)
        // ALL loads of a round before the first use: the round is one memory round trip, not one per 8 records (that form took 5 us per
        // step, all of it latency).  32 records per lane and round; a cell rarely holds more.
        for (int base = 0; base < maxn; base += 32) {         // (wave-uniform trip count)
            U4 r[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                r[j] = U4{{0u, 0u, 0u, 0u}};
                if (base + 4 * j < myc) r[j] = *reinterpret_cast<const U4*>(sp + base + 4 * j);      // may read up to 3 slots past the count: inside the slab (+64 B)
            }
            const int lim = min(maxn - base, 32);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (4 * j < lim) {                            // (wave-uniform)
                    uint2 tx[4], ty[4];
                    bool ok[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t v = r[j].e[k];
                        const int ys = (int)((v >> 12) & 0xFFF);
                        ok[k] = base + 4 * j + k < myc && ys >= ya && ys < yb;
                        tx[k] = xT[ok[k] ? (v & 0xFFF) : 0u]; ty[k] = yT[ok[k] ? ys - ya : 0];
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int hi = ok[k] ? (int)(tx[k].x + ty[k].x) : nh + lane;
                        atomicAdd(&hC[hi], 1);
                        if (rank24) atomicMax(&hB[hi], (r[j].e[k] & 0xFF000000u) | (0xFFFFFFu - (tx[k].y + ty[k].y)));
                    }
                }
            }
        }
    }
    __syncthreads();
    uint2* gl = P.qtLeaf + ((int64_t)frame * P.nlevels + l) * PG_QT_LEAF_CAP;
    const uint32_t cmask = (1u << D) - 1;
    for (int i = tid; i < nh; i += QTP_T) {
        const uint32_t rl = (uint32_t)i / (uint32_t)ncol, col = (uint32_t)i - rl * (uint32_t)ncol, row = (uint32_t)r0 + rl;
        const uint32_t leaf = ((col >> D) << (2 * D)) + qt_spread_bits(col & cmask) + 2u * qt_spread_bits(row);
        gl[leaf] = make_uint2((uint32_t)hC[i], hB[i]);
    }
}

void pg_launch_quadtree(const PgPlan& P, int nframes, hipStream_t s) { pg_launch_quadtree_levels(P, nframes, 0, P.nlevels, s); }

// K3 for the levels [levelBeg, levelEnd) only: a (frame, level) problem depends on nothing but K2's slots of that level
void pg_launch_quadtree_levels(const PgPlan& P, int nframes, int levelBeg, int levelEnd, hipStream_t s)
{
    // dynamic LDS: [node arrays, 28 ints per node (LDS-node levels only) / the prologue's coordinate tables] [aux: max(leaves of the count pyramid, cells + 1) ints]
    int nodes[2] = {0, 0}, aux[2] = {0, 0};
    bool any[2] = {false, false};
    for (int l = levelBeg; l < levelEnd; l++) {
        const int big = P.lvl[l].nodeOff >= 0 ? 1 : 0;
        const int cells = P.lvl[l].nCols * P.lvl[l].nRows + 1;
        const int nIni = max(P.lvl[l].nIni, 1);
        const int leaves = nIni << (2 * qt_pyr_depth(nIni));
        any[big] = true;
        aux[big] = max(aux[big], max(cells, leaves));
        // (the prologue's coordinate tables, 2 ints per region column and row, use the node area before the nodes do)
        nodes[big] = max(nodes[big], 2 * ((P.lvl[l].w - 2 * PG_EDGE) + (P.lvl[l].h - 2 * PG_EDGE)));
        if (!big) nodes[0] = max(nodes[0], P.lvl[l].nodeCap * 28);
    }
    // ONE running maximum per device for all eight instantiations: the attribute is set on every k_quadtree<B, W> at once, so a
    // per-class record would let the class with the smaller need LOWER the limit the other class had just raised
    static size_t configuredDev[64] = {0};                    // (the attribute is per device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& configured = configuredDev[dev & 63];
    // Two launches or one?  (PgPlan::qtSplit: 0 = one, 1 = two, 2 = by this rule.)  Measured on an MI355X, K3 per step, one launch /
    // two launches (tools/experiments/r4_split_grid.sh, profiles/r04_k3_split_grid.txt): 1080p x 1 frame 45 / 40 us, x 16 47 / 46, x 32
    // 52 / 56, x 128 100 / 129; 2160p x 1 99 / 52, x 8 100 / 61, x 32 107 / 107, x 64 139 / 184; 640x480 and 720p: one launch wins at
    // every batch.  The pass kernel is a pure throughput cost (one lane per cell: ~30 % of its lanes carry a record), while inside
    // k_quadtree the pass fills the stalls of the neighbouring workgroups' barriers -- so two launches pay only when the chip is NOT
    // full of quadtree workgroups: a few frames, the more the larger they are.
    const int cells0 = P.lvl[levelBeg].nCols * P.lvl[levelBeg].nRows;
    const bool split = P.qtSplit == 1 || (P.qtSplit == 2 && cells0 >= 1500 && nframes <= 16 + (cells0 - 2304) * 12 / 6912);
    if (split) {
        int groups = 0;
        size_t lds = 0;
        for (int l = levelBeg; l < levelEnd; l++) {
            const PgLevel& V = P.lvl[l];
            const int D = qt_pyr_depth(V.nIni), ncells = V.nCols * V.nRows;
            groups += qt_pass_groups(ncells, D);
            // x table + (at most) the whole y table + the owned rows' counts and best records
            lds = std::max(lds, (size_t)((V.w - 2 * PG_EDGE) + (V.h - 2 * PG_EDGE)) * 8 + (size_t)qt_pass_rows_per_group(ncells, D) * (std::max(V.nIni, 0) << D) * 8 + 2 * 64 * 4);
        }
        static size_t passConfigured[64] = {0};
        int dev0 = 0;
        (void)hipGetDevice(&dev0);
        if (lds > 48 * 1024 && lds > passConfigured[dev0 & 63]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_qt_leaves), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            passConfigured[dev0 & 63] = lds;
        }
        if (groups > 0)
            hipLaunchKernelGGL(k_qt_leaves, dim3(groups, nframes), dim3(QTP_T), lds, s, P, levelBeg, levelEnd - levelBeg);
    }
    for (int big = 0; big < 2; big++) {
        if (!any[big]) continue;
        const size_t lds = (size_t)(nodes[big] + aux[big]) * sizeof(int);
        // Threads per workgroup (profiles/r04_k3_threads_grid.txt, K3 ms per step with 256 / 512 / 1 024 threads): 1080p x 128 frames
        // 0.117 / 0.094 / 0.101, 640 x 480 x 512 0.166 / 0.163 / 0.259, 720p x 128 0.085 / 0.076 / 0.087 -- but one 1080p frame 77 / 52 / 46 us,
        // 1080p x 64 0.089 / 0.066 / 0.063, 2160p x 32 0.249 / 0.152 / 0.113, 1080p / 4000 features x 128 0.219 / 0.170 / 0.164.  Barriers among 8
        // waves are cheaper than among 16 and small problems fit three or four to a CU; a long problem that has a CU to itself wants
        // every wave it can get.  So: 1 024 when the LDS need leaves room for one workgroup per CU anyway; otherwise 512 when the
        // level-0 problem has fewer cells than 1 024 threads could take one each, or when the launch has more problems than the chip
        // has CUs (with the 128-VGPR instantiations: 1080p x 48 frames 0.058 / 0.060 ms, x 64 0.059 / 0.063, x 96 0.076 / 0.090 for 512 /
        // 1 024 threads); 1 024 else.
        const size_t ldsAll = lds + sizeof(int) * (QT_PYR_CAP + 3 * (QT_TMAX / 64) + 8);
        int threads = QT_TMAX;
        if (ldsAll <= 80 * 1024 && (cells0 < 1024 || (int64_t)nframes * (levelEnd - levelBeg) > 256)) threads = 512;
        if (P.qtThreads == 256 || P.qtThreads == 512 || P.qtThreads == 1024) threads = P.qtThreads;
        dim3 grid(nframes, levelEnd - levelBeg), block(threads);
        const int64_t problems = (int64_t)nframes * (levelEnd - levelBeg);
        int wide = 0;
        if (P.qtWide && threads == 512 && ldsAll > 54 * 1024) wide = 1;
        if (P.qtWide && threads == 512 && ldsAll > 40 * 1024 && ldsAll <= 54 * 1024) wide = 3;        // three workgroups per CU: six waves per SIMD, 80 VGPRs
        if (P.qtWide && threads == QT_TMAX && (ldsAll > 80 * 1024 || problems <= 256)) wide = 2;
        if (lds > configured) {
#define PG_QT_ATTR(B, W) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree<B, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
            PG_QT_ATTR(true, 0); PG_QT_ATTR(true, 1); PG_QT_ATTR(true, 2); PG_QT_ATTR(true, 3); PG_QT_ATTR(false, 0); PG_QT_ATTR(false, 1); PG_QT_ATTR(false, 2); PG_QT_ATTR(false, 3);
#undef PG_QT_ATTR
            configured = lds;
        }
#define PG_QT_LAUNCH(B, W) hipLaunchKernelGGL((k_quadtree<B, W>), grid, block, lds, s, P, levelBeg, nodes[big], split ? 1 : 0)
        if (big) { if (wide == 1) PG_QT_LAUNCH(true, 1); else if (wide == 2) PG_QT_LAUNCH(true, 2); else if (wide == 3) PG_QT_LAUNCH(true, 3); else PG_QT_LAUNCH(true, 0); }
        else { if (wide == 1) PG_QT_LAUNCH(false, 1); else if (wide == 2) PG_QT_LAUNCH(false, 2); else if (wide == 3) PG_QT_LAUNCH(false, 3); else PG_QT_LAUNCH(false, 0); }
#undef PG_QT_LAUNCH
    }
}

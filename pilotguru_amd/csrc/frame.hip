// frame.hip -- Frame grid (a12) and ORBmatcher::SearchForInitialization (a10) on gfx950.
//
// Restates (thirdparty/orb-slam2):
//   Frame::AssignFeaturesToGrid / PosInGrid   src/Frame.cc:234-249, 386-396
//   Frame::GetFeaturesInArea                  src/Frame.cc:331-384
//   ORBmatcher::SearchForInitialization       src/ORBmatcher.cc:407-522
//   ORBmatcher::ComputeThreeMaxima            src/ORBmatcher.cc:1605-1646
//
// The matcher is sequential over F1's keypoints by construction: whether candidate i2 is
// considered depends on vMatchedDistance[i2], which earlier keypoints wrote (:445-446, :469).
// What does NOT depend on that order is the expensive part: which keypoints of F2 lie in the
// window of vbPrevMatched[i1] (it is only updated after the loop, :516-519) and their Hamming
// distances.  So the work is split:
//   k_sfi_candidates   one wave per (pair, F1 keypoint), all in parallel: lanes gather the grid
//       cells of the window (CSR ranges, wave prefix sum -> candidate list in the reference's
//       (column, row, insertion) order), filter by level and window (Frame.cc:354-376), evaluate
//       one 256-bit distance each and store the survivors in order as (distance << 16 | i2),
//       at most 64 per keypoint (more: the count says "overflow").
//   k_search_for_initialization   one wave per pair walks F1's keypoints that have candidates, in
//       order, with the stored lists prefetched two groups ahead: per keypoint one LDS gather of
//       vMatchedDistance, two wave reductions ("first minimum wins", :448-457: argmin on
//       distance << 16 | list position; second best over the other entries) and the update by one
//       lane -- no global round trip inside the chain (it was four per keypoint, 3.4 us each:
//       1.46 ms per pair; tools/next_tier_bench.py).  Overflowed keypoints are evaluated in place,
//       cell by cell.  The rotation histogram only needs (i1, the i2 it was matched to when pushed):
//       the bins are computed after the loop, in parallel.
// All per-pair state (vMatchedDistance, vnMatches21, vnMatches12) lives in LDS.
#include "pgorb_internal.h"
#include <algorithm>
#include <string.h>
#include <vector>

#define GRID_COLS PGORB_GRID_COLS
#define GRID_ROWS PGORB_GRID_ROWS
#define GRID_CELLS PGORB_GRID_CELLS
#define HISTO_LENGTH 30
#define TH_LOW 50

int pg_ctx_fail(pgorb_ctx* c, int code, const char* msg);
int pg_ctx_stage(pgorb_ctx* c, int which, size_t bytes, void** p);
int pg_ctx_device(pgorb_ctx* c);
int pg_ctx_scratch(pgorb_ctx* c, size_t bytes, hipStream_t s, void** p);      // the matchers' shared arena, ordered across caller streams
int pg_ctx_scratch_done(pgorb_ctx* c, hipStream_t s);
int pg_ctx_pinned(pgorb_ctx* c, size_t bytes, void** p);                       // the context's page-locked bounce buffer (synchronous host calls only)

__device__ __forceinline__ int grid_cell_of(const pgorb_keypoint& kp, float minX, float minY, float invW, float invH)
{
    const int posX = (int)roundf(__fmul_rn(__fsub_rn(kp.x, minX), invW));       // PosInGrid (:388-389)
    const int posY = (int)roundf(__fmul_rn(__fsub_rn(kp.y, minY), invH));
    return (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) ? -1 : posX * GRID_ROWS + posY;
}

__device__ __forceinline__ int wave_incl_scan(int x, int lane);
__global__ __launch_bounds__(256) void k_frame_grid(const pgorb_keypoint* __restrict__ kps,
                                                     const int32_t* __restrict__ nper, int cap,
                                                     float minX, float minY, float invW, float invH,
                                                     int32_t* __restrict__ gstart, int32_t* __restrict__ gidx)
{
    __shared__ int cnt[GRID_CELLS];
    __shared__ int part[256 + 1];
    __shared__ int cellOf[256];
    const int tid = threadIdx.x, f = blockIdx.x;
    const int n = min(nper[f], cap);
    const pgorb_keypoint* K = kps + (int64_t)f * cap;
    int32_t* start = gstart + (int64_t)f * (GRID_CELLS + 1);
    int32_t* idx = gidx + (int64_t)f * cap;
    for (int c = tid; c < GRID_CELLS; c += 256) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int c = grid_cell_of(K[i], minX, minY, invW, invH);
        if (c >= 0) atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    // exclusive scan of the 3072 counters: 12 per thread
    {
        const int b = tid * 12;
        int sum = 0;
        for (int c = b; c < b + 12; c++) sum += cnt[c];
        // exclusive scan of the 256 partial sums: DPP scan inside each wave, the four wave totals by every thread (one thread walking
        // the 256 entries cost 14 of the kernel's 32 us)
        const int lane_ = tid & 63, wv_ = tid >> 6;
        const int incl = wave_incl_scan(sum, lane_);
        if (lane_ == 63) part[wv_] = incl;
        __syncthreads();
        int base_ = 0, total_ = 0;
        for (int w = 0; w < 4; w++) { const int v = part[w]; base_ += w < wv_ ? v : 0; total_ += v; }
        int run = base_ + incl - sum;
        if (tid == 255) part[256] = total_;
        for (int c = b; c < b + 12; c++) { const int v = cnt[c]; cnt[c] = run; start[c] = run; run += v; }
        if (tid == 255) start[GRID_CELLS] = total_;
    }
    __syncthreads();
    // stable placement: chunks of 256 keypoints in index order (mGrid[..].push_back(i), :246-247)
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        const int c = (i < n) ? grid_cell_of(K[i], minX, minY, invW, invH) : -1;
        cellOf[tid] = c;
        __syncthreads();
        int pos = -1;
        if (c >= 0) {
            int rank = 0;
            for (int t = 0; t < tid; t++) rank += (cellOf[t] == c);
            pos = cnt[c] + rank;
        }
        __syncthreads();
        if (c >= 0) { atomicAdd(&cnt[c], 1); idx[pos] = i; }
        __syncthreads();
    }
}

// wave-wide inclusive sum / minimum on DPP row shifts and broadcasts (6 cross-lane moves on the VALU; the __shfl forms
// go through the LDS crossbar, ~100 cycles each, and the sequential matcher pass pays every one of them in full)
__device__ __forceinline__ int wave_incl_scan(int x, int lane)
{
    (void)lane;
    int v = x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x)
{
    int v = (int)x;                                                      // lanes a shift does not reach keep their own value
    v = (int)min((unsigned)v, (unsigned)__builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = (int)min((unsigned)v, (unsigned)__builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = (int)min((unsigned)v, (unsigned)__builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = (int)min((unsigned)v, (unsigned)__builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));   // lane 15 of every row: the row's minimum
    v = (int)min((unsigned)v, (unsigned)__builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = (int)min((unsigned)v, (unsigned)__builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}

// smallest and second-smallest of the lanes' (distinct or 0xFFFFFFFF) keys in ONE pass: every step merges two (min, second) pairs
__device__ __forceinline__ void wave_min2_u32(unsigned x, unsigned& best, unsigned& second)
{
    int a = (int)x, b = -1;                                              // (min, second) of the lanes seen so far; -1 = 0xFFFFFFFF
#define PG_MIN2_STEP(CTRL, ROWMASK) do { \
        const unsigned oa = (unsigned)__builtin_amdgcn_update_dpp(-1, a, CTRL, ROWMASK, 0xf, false); \
        const unsigned ob = (unsigned)__builtin_amdgcn_update_dpp(-1, b, CTRL, ROWMASK, 0xf, false); \
        const unsigned hi = max((unsigned)a, oa); \
        a = (int)min((unsigned)a, oa); \
        b = (int)min(min((unsigned)b, ob), hi); } while (0)
    PG_MIN2_STEP(0x111, 0xf); PG_MIN2_STEP(0x112, 0xf); PG_MIN2_STEP(0x114, 0xf); PG_MIN2_STEP(0x118, 0xf);
    PG_MIN2_STEP(0x142, 0xa); PG_MIN2_STEP(0x143, 0xc);
#undef PG_MIN2_STEP
    best = (unsigned)__builtin_amdgcn_readlane(a, 63);
    second = (unsigned)__builtin_amdgcn_readlane(b, 63);
}

// vbPrevMatched of MonocularInitialization: the reference frame's keypoint positions (Tracking.cc:583-585)
__global__ __launch_bounds__(256) void k_prev_matched_init(const pgorb_keypoint* __restrict__ kps, int64_t rows, float2* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows) out[i] = make_float2(kps[i].x, kps[i].y);
}
void pg_launch_prev_matched_init(const pgorb_keypoint* d_kps, int64_t rows, float* d_out, hipStream_t s)
{
    if (rows > 0) hipLaunchKernelGGL(k_prev_matched_init, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, d_kps, rows, reinterpret_cast<float2*>(d_out));
}

extern __shared__ __attribute__((aligned(16))) uint8_t pg_sfi_smem[];

// ---- candidate lists of the two-pass matchers (round 4: variable length) -----------------------------------------------
// Pass A stores EVERY surviving candidate of a query in the reference's scan order: the first LIST_K in the query's fixed slots,
// the rest in the pair's pool (one atomic per query that needs it).  Rounds 2-3 capped the lists at 64 and re-evaluated denser
// queries in place inside the sequential pass -- the initialisation workload's cliff.  Only when a pair's pool is full (an
// average of LIST_K + LIST_POOL candidates per query) is a query still evaluated in place (count LIST_OVER).
#define LIST_K 64
#define LIST_POOL 256
#define LIST_OVER 0xFFFFu
#define SFI_K LIST_K
struct PgLists {
    uint32_t* fixed;             // [rows][LIST_K]
    uint16_t* cnt;               // [rows]   survivors of the query (LIST_OVER: evaluate in place)
    uint32_t* ovf;               // [rows]   where the query's entries LIST_K.. start in its pair's pool
    uint32_t* pool;              // [npairs][poolPerPair]
    int32_t*  poolTop;           // [npairs] (zeroed before pass A)
    uint32_t  poolPerPair;
};
// scratch layout for npairs x rowsPerPair rows; returns the bytes needed
static size_t pg_lists_layout(void* scratch, int npairs, int rowsPerPair, PgLists* L)
{
    const size_t rows = (size_t)npairs * rowsPerPair;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    uint8_t* b = (uint8_t*)scratch;
    L->poolTop = (int32_t*)(b + off); off += al((size_t)npairs * 4);
    L->fixed = (uint32_t*)(b + off); off += al(rows * LIST_K * 4);
    L->cnt = (uint16_t*)(b + off); off += al(rows * 2);
    L->ovf = (uint32_t*)(b + off); off += al(rows * 4);
    L->poolPerPair = (uint32_t)((size_t)rowsPerPair * LIST_POOL);
    L->pool = (uint32_t*)(b + off); off += al((size_t)npairs * L->poolPerPair * 4);
    return off;
}
// entry `pos` (any position) of a query row; chunk = 64 consecutive entries, one per lane
__device__ __forceinline__ uint32_t pg_list_chunk(const PgLists& L, int64_t row, int p, uint32_t ovf, int ch, int lane)
{
    return ch == 0 ? L.fixed[row * LIST_K + lane] : L.pool[(size_t)p * L.poolPerPair + ovf + (uint32_t)(ch - 1) * 64u + (uint32_t)lane];
}
// Pass A, the tail of a query's wave: `total` survivors are about to be written.  Reserves pool space when they do not fit the
// fixed slots; returns the pool offset (wave-uniform) and sets `over` when the pair's pool is full.
__device__ __forceinline__ uint32_t pg_list_reserve(const PgLists& L, int p, int total, int lane, bool& over)
{
    over = false;
    if (total <= LIST_K) return 0u;
    const int need = (total - LIST_K + 63) & ~63;
    int base = 0;
    if (lane == 0) base = atomicAdd(&L.poolTop[p], need);
    base = __builtin_amdgcn_readfirstlane(base);
    over = (uint32_t)base + (uint32_t)need > L.poolPerPair;
    return (uint32_t)base;
}

// GetFeaturesInArea's cell window (Frame.cc:336-350); false = the reference returns an empty vector
__device__ __forceinline__ bool sfi_window(float x, float y, float r, float minX, float minY, float invW, float invH,
                                           int& cx0, int& cx1, int& cy0, int& cy1)
{
    cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, minX), r), invW)));
    if (cx0 >= GRID_COLS) return false;
    cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, minX), r), invW)));
    if (cx1 < 0) return false;
    cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, minY), r), invH)));
    if (cy0 >= GRID_ROWS) return false;
    cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, minY), r), invH)));
    if (cy1 < 0) return false;
    return cx1 >= cx0 && cy1 >= cy0;
}

__device__ __forceinline__ int sfi_distance(const uint4 q0, const uint4 q1, const uint8_t* d)
{
    const uint4 d0 = reinterpret_cast<const uint4*>(d)[0], d1 = reinterpret_cast<const uint4*>(d)[1];
    return __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
           __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
}

// Phase 1: candidate lists.  Workgroup = 4 waves = 4 consecutive F1 keypoints of pair blockIdx.y.
__global__ __launch_bounds__(256) void k_sfi_candidates(
    const pgorb_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc, const int32_t* __restrict__ nper,
    int cap, const int32_t* __restrict__ gstart, const int32_t* __restrict__ gidx,
    const int32_t* __restrict__ pairF1, const int32_t* __restrict__ pairF2,
    float minX, float minY, float invW, float invH, const float* __restrict__ prevMatched, int windowSize, PgLists Ls)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, p = blockIdx.y;
    const int i1 = blockIdx.x * 4 + wv;
    const int f1 = pairF1[p], f2 = pairF2[p];
    const int n1 = min(nper[f1], cap);
    if (i1 >= n1) return;
    const int64_t row = (int64_t)p * cap + i1;
    uint16_t* cntOut = Ls.cnt + row;
    const pgorb_keypoint kp1 = kps[(int64_t)f1 * cap + i1];
    int cx0, cx1, cy0, cy1;
    const float x = prevMatched[((int64_t)p * cap + i1) * 2], y = prevMatched[((int64_t)p * cap + i1) * 2 + 1];
    const float r = (float)windowSize;
    if (kp1.octave > 0 || !sfi_window(x, y, r, minX, minY, invW, invH, cx0, cx1, cy0, cy1)) {        // :424-426
        if (lane == 0) *cntOut = 0;
        return;
    }
    const int level1 = kp1.octave;
    const pgorb_keypoint* K2 = kps + (int64_t)f2 * cap;
    const uint8_t* D2 = desc + (int64_t)f2 * cap * 32;
    const int32_t* start2 = gstart + (int64_t)f2 * (GRID_CELLS + 1);
    const int32_t* idx2 = gidx + (int64_t)f2 * cap;
    uint16_t* candList = reinterpret_cast<uint16_t*>(pg_sfi_smem) + (size_t)wv * cap;      // this wave's vIndices2 before filtering
    const int ncy = cy1 - cy0 + 1, T = (cx1 - cx0 + 1) * ncy;
    int M = 0;
    for (int base = 0; base < T; base += 64) {                  // window cells in (ix, iy) order, entries in insertion order
        const int t = base + lane;
        int s0 = 0, cnt = 0;
        if (t < T) {
            const int c = (cx0 + t / ncy) * GRID_ROWS + cy0 + t % ncy;
            s0 = start2[c]; cnt = start2[c + 1] - s0;
        }
        const int incl = wave_incl_scan(cnt, lane);
        const int off = M + incl - cnt;
        for (int j = 0; j < cnt; j++) candList[off + j] = (uint16_t)idx2[s0 + j];
        M += __builtin_amdgcn_readlane(incl, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint4 q0 = reinterpret_cast<const uint4*>(desc + ((int64_t)f1 * cap + i1) * 32)[0];
    const uint4 q1 = reinterpret_cast<const uint4*>(desc + ((int64_t)f1 * cap + i1) * 32)[1];
    // bCheckLevels is true for minLevel = maxLevel = 0 (Frame.cc:354): octave must equal level1
    auto survives = [&](int k, int& i2) {
        i2 = candList[k];
        const pgorb_keypoint kp2 = K2[i2];
        return kp2.octave == level1 && fabsf(__fsub_rn(kp2.x, x)) < r && fabsf(__fsub_rn(kp2.y, y)) < r;
    };
    // survivors beyond the fixed slots go to the pair's pool, reserved in one piece the moment the 65th survivor turns up -- for what
    // is left of the window's M keypoints, an upper bound (counting the survivors first cost a second pass over the keypoints, and
    // reserving for every query with M > 64 an atomic per query on the pair's counter: + 25 % / + 100 % on the whole matcher)
    int total = 0;
    bool over = false, reserved = false;
    uint32_t ovf = 0;
    uint32_t* out = Ls.fixed + row * LIST_K;
    uint32_t* outPool = Ls.pool + (size_t)p * Ls.poolPerPair;
    for (int base = 0; base < M; base += 64) {
        const int k = base + lane;
        int i2 = 0;
        const bool ok = k < M && survives(k, i2);
        const uint32_t e = ok ? (((uint32_t)sfi_distance(q0, q1, D2 + (int64_t)i2 * 32) << 16) | (uint32_t)i2) : 0u;
        const unsigned long long m = __ballot(ok);
        const int pos = total + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        total += __popcll(m);
        if (total > LIST_K && !reserved) { ovf = pg_list_reserve(Ls, p, LIST_K + (M - base), lane, over); reserved = true; }     // (wave-uniform)
        if (ok) { if (pos < LIST_K) out[pos] = e; else if (!over) outPool[ovf + (uint32_t)(pos - LIST_K)] = e; }
    }
    if (lane == 0) { *cntOut = (uint16_t)(over ? LIST_OVER : total); Ls.ovf[row] = ovf; }
}

// A keypoint of F1 with more than SFI_K candidates in its window (rare: every keypoint of a dense patch within 100 px):
// distances and the vMatchedDistance filter evaluated in place, cell by cell in the reference's (column, row, insertion)
// order.  Out: smallest (distance << 16 | running position), the second-smallest distance, the winner's i2.
// (results by value: reference parameters of a non-inlined function live in scratch memory, and the common path paid for it)
__device__ __noinline__ uint3 sfi_eval_in_place(const pgorb_keypoint kp1, float x, float y, float r, float minX, float minY, float invW,
                                                float invH, const uint8_t* d1, const pgorb_keypoint* K2, const uint8_t* D2,
                                                const int32_t* start2, const int32_t* idx2, const uint16_t* matchedDist, int lane)
{
    int cx0, cx1, cy0, cy1;
    sfi_window(x, y, r, minX, minY, invW, invH, cx0, cx1, cy0, cy1);          // (true: phase 1 got here)
    const uint4 q0 = reinterpret_cast<const uint4*>(d1)[0], q1 = reinterpret_cast<const uint4*>(d1)[1];
    unsigned b1key = 0xFFFFFFFFu, b1idx = 0; int b2 = 0x7fffffff; int posBase = 0;
    for (int ix = cx0; ix <= cx1; ix++)
        for (int iy = cy0; iy <= cy1; iy++) {
            const int c = ix * GRID_ROWS + iy, s0 = start2[c], cnt = start2[c + 1] - s0;
            for (int k = lane; k < cnt; k += 64) {
                const int i2 = idx2[s0 + k];
                const pgorb_keypoint kp2 = K2[i2];
                const float distx = __fsub_rn(kp2.x, x), disty = __fsub_rn(kp2.y, y);
                if (kp2.octave != kp1.octave || !(fabsf(distx) < r && fabsf(disty) < r)) continue;
                const int dist = sfi_distance(q0, q1, D2 + (int64_t)i2 * 32);
                if ((int)matchedDist[i2] <= dist) continue;
                const unsigned key = ((unsigned)dist << 16) | (unsigned)(posBase + k);     // posBase + k < cap < 2^16
                if (key < b1key) { if (b1key != 0xFFFFFFFFu) b2 = min(b2, (int)(b1key >> 16)); b1key = key; b1idx = (unsigned)i2; }
                else b2 = min(b2, dist);
            }
            posBase += cnt;
        }
    const unsigned wkey = wave_min_u32(b1key);
    if (wkey == 0xFFFFFFFFu) return make_uint3(wkey, 0x7fffffffu, 0u);
    const unsigned long long who = __ballot(b1key == wkey);
    const int bestIdx2 = __shfl((int)b1idx, __ffsll((long long)who) - 1);
    const unsigned mine = (b1key == wkey) ? (unsigned)b2 : (b1key == 0xFFFFFFFFu ? 0x7fffffffu : (b1key >> 16));
    return make_uint3(wkey, wave_min_u32(min(mine, (unsigned)b2)), (unsigned)bestIdx2);
}

// The long forms of a keypoint's evaluation in the sequential pass, out of line: a list longer than the fixed slots (chunk 0 = the
// prefetched `e0`, the rest 64 entries at a time from the pair's pool), or -- the pair's pool was full -- the evaluation in place.
// Out: smallest (distance << 16 | position), the second-smallest distance (0x7fffffff: none), the winner's i2.
__device__ __noinline__ uint3 sfi_eval_long(int count, uint32_t e0, const uint32_t* poolRow, const pgorb_keypoint kp1, float x, float y, float r,
                                            float minX, float minY, float invW, float invH, const uint8_t* d1, const pgorb_keypoint* K2,
                                            const uint8_t* D2, const int32_t* start2, const int32_t* idx2, const uint16_t* matchedDist, int lane)
{
    if (count == (int)LIST_OVER) return sfi_eval_in_place(kp1, x, y, r, minX, minY, invW, invH, d1, K2, D2, start2, idx2, matchedDist, lane);
    unsigned wkey = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
    int bestIdx2 = -1;
    for (int ch = 0; ch * 64 < count; ch++) {
        const uint32_t ee = ch == 0 ? e0 : poolRow[(uint32_t)(ch - 1) * 64u + (uint32_t)lane];
        const int i2 = (int)(ee & 0xFFFFu), dist = (int)(ee >> 16);
        const bool keep = ch * 64 + lane < count && !((int)matchedDist[i2] <= dist);       // :445-446
        const unsigned key = keep ? (((unsigned)dist << 16) | (unsigned)(ch * 64 + lane)) : 0xFFFFFFFFu;
        unsigned k1, k2;
        wave_min2_u32(key, k1, k2);
        if (k1 < wkey) { second = min(wkey, k2); wkey = k1; bestIdx2 = __builtin_amdgcn_readlane(i2, (int)(k1 & 63u)); }
        else second = min(second, k1);
    }
    return make_uint3(wkey, second == 0xFFFFFFFFu ? 0x7fffffffu : (second >> 16), (unsigned)bestIdx2);
}

#define SFI_G 8                  // keypoints per prefetch group
// Phase 2: the sequential pass, one wave per pair.  (Round 4 also built this pass as ROUNDS of independent keypoints -- the scheme
// k_search_by_projection runs below -- and measured it slower here: every listed keypoint of a pair is a level-0 keypoint with a
// 200-px window, the lists overlap heavily, and the conservative readiness rule left ~10 % of the keypoints per round: 1.33 ms per 127
// pairs of 4 000 features against 0.68 ms for this walk; profiles/r04_next_tier.txt.  The lists are variable length now: a dense
// window no longer falls back to the evaluation in place.)
__global__ __launch_bounds__(64) void k_search_for_initialization(
    const pgorb_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc, const int32_t* __restrict__ nper,
    int cap, const int32_t* __restrict__ gstart, const int32_t* __restrict__ gidx,
    const int32_t* __restrict__ pairF1, const int32_t* __restrict__ pairF2,
    float minX, float minY, float invW, float invH,
    float* __restrict__ prevMatched, int32_t* __restrict__ matches12out, int32_t* __restrict__ nmatchesOut,
    int windowSize, float nnratio, int checkOrientation, PgLists Ls)
{
    const int lane = threadIdx.x, p = blockIdx.x;
    const int f1 = pairF1[p], f2 = pairF2[p];
    const int n1 = min(nper[f1], cap);
    const pgorb_keypoint* K1 = kps + (int64_t)f1 * cap;
    const pgorb_keypoint* K2 = kps + (int64_t)f2 * cap;
    const uint8_t* D1 = desc + (int64_t)f1 * cap * 32;
    const uint8_t* D2 = desc + (int64_t)f2 * cap * 32;
    const int32_t* start2 = gstart + (int64_t)f2 * (GRID_CELLS + 1);
    const int32_t* idx2 = gidx + (int64_t)f2 * cap;
    float* prev = prevMatched + (int64_t)p * cap * 2;
    int32_t* m12out = matches12out + (int64_t)p * cap;
    const int64_t row0 = (int64_t)p * cap;
    const uint32_t* L = Ls.fixed + row0 * LIST_K;
    const uint16_t* LC = Ls.cnt + row0;

    uint16_t* matchedDist = reinterpret_cast<uint16_t*>(pg_sfi_smem);      // [cap] vMatchedDistance (0xFFFF = INT_MAX)
    int16_t* m21 = reinterpret_cast<int16_t*>(matchedDist + cap);          // [cap] vnMatches21
    int16_t* m12 = m21 + cap;                                              // [cap] vnMatches12
    int16_t* push2 = m12 + cap;                                            // [cap] i2 an i1 was matched to when it entered the histogram, or -1
    uint16_t* active = reinterpret_cast<uint16_t*>(push2 + cap);           // [cap] F1 keypoints with candidates, in order
    for (int i = lane; i < cap; i += 64) { matchedDist[i] = 0xFFFF; m21[i] = -1; m12[i] = -1; push2[i] = -1; }
    int nact = 0;
    for (int base = 0; base < n1; base += 512) {                             // (8 count loads in flight, not one round trip per 64 keypoints)
        uint16_t cv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + 64 * u + lane; cv[u] = i < n1 ? LC[i] : (uint16_t)0; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool on = cv[u] != 0;
            const unsigned long long m = __ballot(on);
            if (on) active[nact + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = (uint16_t)(base + 64 * u + lane);
            nact += __popcll(m);
        }
    }
    __syncthreads();

    const float r = (float)windowSize;
    int nmatches = 0;
    // group g = active[g * SFI_G .. ): entry `lane` of each member's list and its count, loaded one group ahead
    uint32_t curE[SFI_G], nxtE[SFI_G]; int curC[SFI_G], nxtC[SFI_G], curI[SFI_G], nxtI[SFI_G];
    auto load_group = [&](int g, uint32_t (&E)[SFI_G], int (&Cn)[SFI_G], int (&I)[SFI_G]) {
#pragma unroll
        for (int j = 0; j < SFI_G; j++) {
            const int a = g * SFI_G + j;
            I[j] = -1; Cn[j] = 0; E[j] = 0;
            if (a < nact) {
                const int i1 = active[a];
                I[j] = i1; Cn[j] = LC[i1];
                E[j] = L[(int64_t)i1 * SFI_K + lane];                        // (all 64 slots: no wait for the count; slots past it are masked below)
            }
        }
    };
    const int ngroups = (nact + SFI_G - 1) / SFI_G;
    if (ngroups) load_group(0, curE, curC, curI);
    for (int g = 0; g < ngroups; g++) {
        if (g + 1 < ngroups) load_group(g + 1, nxtE, nxtC, nxtI);
#pragma unroll
        for (int j = 0; j < SFI_G; j++) {
            const int i1 = curI[j];
            if (i1 < 0) break;                                              // (wave-uniform)
            unsigned wkey; int bestIdx2 = -1; unsigned second;
            if (curC[j] <= LIST_K) {
                // the common case, straight: the whole list is the prefetched chunk
                const int i2 = (int)(curE[j] & 0xFFFFu), dist = (int)(curE[j] >> 16);
                const bool keep = lane < curC[j] && !((int)matchedDist[i2] <= dist);       // :445-446
                const unsigned key = keep ? (((unsigned)dist << 16) | (unsigned)lane) : 0xFFFFFFFFu;
                wave_min2_u32(key, wkey, second);                           // smallest key, and the smallest of the others
                if (wkey == 0xFFFFFFFFu) continue;
                bestIdx2 = __builtin_amdgcn_readlane(i2, (int)(wkey & 0xFFFFu));
                second = (second == 0xFFFFFFFFu) ? 0x7fffffffu : (second >> 16);
            } else {
                // a dense window (the list continues in the pair's pool) or a pair whose pool is full (evaluation in place): out of line, so
                // that the eight unrolled copies of this body stay small -- inlined, the sequential wave lost 20 % to instruction fetch
                const uint3 ev = sfi_eval_long(curC[j], curE[j], Ls.pool + (size_t)p * Ls.poolPerPair + Ls.ovf[row0 + i1], K1[i1], prev[2 * i1], prev[2 * i1 + 1], r,
                                               minX, minY, invW, invH, D1 + (int64_t)i1 * 32, K2, D2, start2, idx2, matchedDist, lane);
                wkey = ev.x; second = ev.y; bestIdx2 = (int)ev.z;
                if (wkey == 0xFFFFFFFFu) continue;
            }
            const int bestDist = (int)(wkey >> 16);
            const float bestDist2 = (second >= 0x7fffffffu) ? 2147483648.0f : (float)(int)second;   // (float)INT_MAX
            if (bestDist <= TH_LOW && (float)bestDist < __fmul_rn(bestDist2, nnratio)) {            // :460-462
                const int old = m21[bestIdx2];
                if (old >= 0) nmatches--;                                    // :464-468
                nmatches++;
                if (lane == 0) {
                    if (old >= 0) m12[old] = -1;
                    m12[i1] = (int16_t)bestIdx2;
                    m21[bestIdx2] = (int16_t)i1;
                    matchedDist[bestIdx2] = (uint16_t)bestDist;
                    push2[i1] = (int16_t)bestIdx2;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
#pragma unroll
        for (int j = 0; j < SFI_G; j++) { curE[j] = nxtE[j]; curC[j] = nxtC[j]; curI[j] = nxtI[j]; }
    }
    __syncthreads();
    if (checkOrientation) {
        // histogram sizes = number of pushes per bin (a displaced i1 stays in its list, :481); the bin of a push is
        // a function of the two keypoints' angles (:473-483)
        const float factor = 1.0f / HISTO_LENGTH;
        int8_t* rotBin = reinterpret_cast<int8_t*>(active);                  // [n1] (the active list is done)
        int* hist = reinterpret_cast<int*>(pg_sfi_smem + (((size_t)cap * 10 + 3) & ~(size_t)3));      // [32] behind the arrays
        if (lane < 32) hist[lane] = 0;
        __syncthreads();
        for (int base = 0; base < n1; base += 256) {                         // (the angle loads of 4 x 64 keypoints in flight)
            int i2v[4]; float a1[4], a2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + 64 * u + lane;
                i2v[u] = i < n1 ? (int)push2[i] : -1;
                a1[u] = 0.f; a2[u] = 0.f;
                if (i2v[u] >= 0) { a1[u] = K1[i].angle; a2[u] = K2[i2v[u]].angle; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + 64 * u + lane;
                int bin = -1;
                if (i2v[u] >= 0) {
                    float rot = __fsub_rn(a1[u], a2[u]);
                    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                    bin = (int)roundf(__fmul_rn(rot, factor));
                    if (bin == HISTO_LENGTH) bin = 0;
                    atomicAdd(&hist[bin], 1);
                }
                if (i < n1) rotBin[i] = (int8_t)bin;
            }
        }
        __syncthreads();
        const int h = lane < HISTO_LENGTH ? hist[lane] : 0;                  // lane b < 30 holds the size of bin b
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {                             // ComputeThreeMaxima (:1605-1646)
            const int s = __shfl(h, i);
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        int removed = 0;
        for (int i = lane; i < n1; i += 64) {
            const int b = rotBin[i];
            if (b >= 0 && b != ind1 && b != ind2 && b != ind3 && m12[i] >= 0) { m12[i] = -1; removed++; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) removed += __shfl_xor(removed, d);
        nmatches -= removed;
        __syncthreads();
    }
    for (int base = 0; base < n1; base += 256) {                             // :516-519
        int mv[4]; float2 xy[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + 64 * u + lane;
            mv[u] = i < n1 ? (int)m12[i] : -1;
            xy[u] = make_float2(0.f, 0.f);
            if (mv[u] >= 0) xy[u] = *reinterpret_cast<const float2*>(&K2[mv[u]].x);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + 64 * u + lane;
            if (i < n1) m12out[i] = mv[u];
            if (mv[u] >= 0) *reinterpret_cast<float2*>(prev + 2 * i) = xy[u];
        }
    }
    if (lane == 0) nmatchesOut[p] = nmatches;
}

#ifndef RR_T
#define RR_T 1024               // threads of the rounds workgroup (one per pair); 512 in a developer build: tools/experiments/r4_rr_threads.sh
#endif
#define RR_W (RR_T / 64)
// ---- SearchByProjection (local map points / last frame), src/ORBmatcher.cc:46-131, 1355-1474 ----
// One wave per frame; queries in order.  mode 0: best + second with the same-level ratio test
// (:83-125); mode 1: best only + rotation histogram (:1390-1469).
// Batch layout (round 3): pair p = blockIdx.x matches its nq[p] queries against frame pairFrame[p] of an extract batch
// (keypoints / descriptors `cap` apart, grids (GRID_CELLS + 1) / cap apart); query arrays are [npairs][qcap].  The search
// radius and the level window of a query are derived here from what the caller holds (predicted level + viewing cosine,
// or the last frame's octave) exactly as the reference does, so the host never touches the queries.
struct PgProjBatch {
    const pgorb_keypoint* K; const uint8_t* D; const int32_t* n; int cap;
    const int32_t* gstart; const int32_t* gidx; const int32_t* pairFrame;
    const uint8_t* kpHasPoint;             // [npairs][cap] or null
    int qcap; const int32_t* nq;
    const uint8_t* valid; const float* x; const float* y; const int32_t* level; const float* aux;   // aux: view cos (mode 0) / angle (mode 1)
    const uint8_t* desc; const uint8_t* hasObs;
    float sf[PG_MAXL + 1]; int nlevels; float th;
    // mode 2 (key frame, relocalisation): level = PredictScale(dist3d), aux = the key frame keypoint's angle
    const uint8_t* found; const float* dist3d; const float* minDist; const float* maxDist; float logSf; int orbDist;
    float maxX, maxY;                      // mnMaxX / mnMaxY (the kernels derive everything else from minX / minY and the inverse cell sizes)
};

#define TH_HIGH 100

// Round 3, two passes like SearchForInitialization: the candidates of a query and their distances do not depend on the
// assignments made so far (only `taken` does), so pass A computes them for every query of every pair in parallel and
// pass B -- one wave per pair, the reference's order -- only filters the stored candidates by `taken` and picks.
#define PROJ_K LIST_K            // candidates in a query's fixed slots; the rest of its list is in the pair's pool (PgLists)

// GetFeaturesInArea's window and the level range of query q; false = the reference skips the query
__device__ __forceinline__ bool proj_query(const PgProjBatch& B, int64_t qi, int mode, float minX, float minY, float invW, float invH,
                                           float& x, float& y, float& r, int& minLevel, int& maxLevel, int& cx0, int& cx1, int& cy0, int& cy1)
{
    if (!B.valid[qi]) return false;
    x = B.x[qi]; y = B.y[qi];
    int lvl;
    if (mode == 2) {
        // ORBmatcher.cc:1497-1531: not already found, projection inside the image bounds, depth inside the point's scale
        // invariance range, level from MapPoint::PredictScale
        if (B.found[qi]) return false;
        if (x < minX || x > B.maxX || y < minY || y > B.maxY) return false;          // :1512-1515
        const float d3 = B.dist3d[qi], dmin = B.minDist[qi], dmax = B.maxDist[qi];
        if (d3 < dmin || d3 > dmax) return false;
        lvl = pg_predict_scale(dmax, d3, B.logSf, B.nlevels);
    } else {
        lvl = B.level[qi];
        if (lvl < 0 || lvl >= B.nlevels) return false;
    }
    if (mode == 2) {
        r = __fmul_rn(B.th, B.sf[lvl]);                               // th * CurrentFrame.mvScaleFactors[nPredictedLevel] (:1531)
        minLevel = lvl - 1; maxLevel = lvl + 1;                       // :1533
    } else if (mode == 0) {
        r = ((double)B.aux[qi] > 0.998) ? 2.5f : 4.0f;                // RadiusByViewingCos (:133-139)
        if (B.th != 1.0f) r = __fmul_rn(r, B.th);                     // bFactor (:50, :65-66)
        r = __fmul_rn(r, B.sf[lvl]);                                  // r * F.mvScaleFactors[nPredictedLevel] (:69)
        minLevel = lvl - 1; maxLevel = lvl;                           // :69-70
    } else {
        r = __fmul_rn(B.th, B.sf[lvl]);                               // th * CurrentFrame.mvScaleFactors[nLastOctave] (:1383)
        minLevel = lvl - 1; maxLevel = lvl + 1;                       // :1392 (mono: neither forward nor backward)
    }
    return sfi_window(x, y, r, minX, minY, invW, invH, cx0, cx1, cy0, cy1);   // Frame.cc:336-350
}

// entry of a stored candidate: distance << 23 | rotation bin << 18 | octave << 14 | keypoint index
__device__ __forceinline__ uint32_t proj_entry(int dist, int bin, int octave, int i2)
{
    return ((uint32_t)dist << 23) | ((uint32_t)(bin & 31) << 18) | ((uint32_t)(octave & 15) << 14) | (uint32_t)i2;
}
__device__ __forceinline__ int proj_bin(float qangle, float kangle)
{
    float rot = __fsub_rn(qangle, kangle);                            // :1428-1434
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// Pass A: workgroup = 4 waves = 4 consecutive queries of pair blockIdx.y
__global__ __launch_bounds__(256) void k_proj_candidates(PgProjBatch B, float minX, float minY, float invW, float invH, int mode, PgLists Ls)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, p = blockIdx.y;
    const int q = blockIdx.x * 4 + wv;
    const int nq = min(B.nq[p], B.qcap);
    if (q >= nq) return;
    const int frame = B.pairFrame ? B.pairFrame[p] : p, cap = B.cap;
    const int64_t qi = (int64_t)p * B.qcap + q;
    uint16_t* cntOut = Ls.cnt + qi;
    float x, y, r; int minLevel, maxLevel, cx0, cx1, cy0, cy1;
    if (!proj_query(B, qi, mode, minX, minY, invW, invH, x, y, r, minLevel, maxLevel, cx0, cx1, cy0, cy1)) {
        if (lane == 0) *cntOut = 0;
        return;
    }
    const pgorb_keypoint* __restrict__ K = B.K + (int64_t)frame * cap;
    const uint8_t* __restrict__ D = B.D + (int64_t)frame * cap * 32;
    const int32_t* __restrict__ gstart = B.gstart + (int64_t)frame * (GRID_CELLS + 1);
    const int32_t* __restrict__ gidx = B.gidx + (int64_t)frame * cap;
    uint16_t* candList = reinterpret_cast<uint16_t*>(pg_sfi_smem) + (size_t)wv * cap;
    const int ncy = cy1 - cy0 + 1, T = (cx1 - cx0 + 1) * ncy;
    int M = 0;
    for (int base = 0; base < T; base += 64) {                  // window cells in (ix, iy) order, entries in insertion order
        const int t = base + lane;
        int s0 = 0, cnt = 0;
        if (t < T) {
            const int c = (cx0 + t / ncy) * GRID_ROWS + cy0 + t % ncy;
            s0 = gstart[c]; cnt = gstart[c + 1] - s0;
        }
        const int incl = wave_incl_scan(cnt, lane);
        const int off = M + incl - cnt;
        for (int j = 0; j < cnt; j++) candList[off + j] = (uint16_t)gidx[s0 + j];
        M += __builtin_amdgcn_readlane(incl, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const uint4 q0 = reinterpret_cast<const uint4*>(B.desc + qi * 32)[0];
    const uint4 q1 = reinterpret_cast<const uint4*>(B.desc + qi * 32)[1];
    const float qangle = mode != 0 ? B.aux[qi] : 0.f;
    auto survives = [&](int k, int& i2, pgorb_keypoint& kp2) {
        i2 = candList[k];
        kp2 = K[i2];
        if (bCheckLevels && (kp2.octave < minLevel || (maxLevel >= 0 && kp2.octave > maxLevel))) return false;
        return fabsf(__fsub_rn(kp2.x, x)) < r && fabsf(__fsub_rn(kp2.y, y)) < r;
    };
    // (survivors beyond the fixed slots: the pair's pool, reserved when the 65th turns up -- see k_sfi_candidates)
    int total = 0;
    bool over = false, reserved = false;
    uint32_t ovf = 0;
    uint32_t* out = Ls.fixed + qi * LIST_K;
    uint32_t* outPool = Ls.pool + (size_t)p * Ls.poolPerPair;
    for (int base = 0; base < M; base += 64) {
        const int k = base + lane;
        int i2 = 0; pgorb_keypoint kp2;
        const bool ok = k < M && survives(k, i2, kp2);
        const uint32_t e = ok ? proj_entry(sfi_distance(q0, q1, D + (int64_t)i2 * 32), mode != 0 ? proj_bin(qangle, kp2.angle) : 0, kp2.octave, i2) : 0u;
        const unsigned long long m = __ballot(ok);
        const int pos = total + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        total += __popcll(m);
        if (total > LIST_K && !reserved) { ovf = pg_list_reserve(Ls, p, LIST_K + (M - base), lane, over); reserved = true; }     // (wave-uniform)
        if (ok) { if (pos < LIST_K) out[pos] = e; else if (!over) outPool[ovf + (uint32_t)(pos - LIST_K)] = e; }
    }
    if (lane == 0) { *cntOut = (uint16_t)(over ? LIST_OVER : total); Ls.ovf[qi] = ovf; }
}

// a query with more than PROJ_K candidates: the whole evaluation in place, in the reference's order (the round-2 form of the
// kernel); returns the best two entries, their keys' distances in the entry's distance field
__device__ __noinline__ uint2 proj_eval_in_place(const PgProjBatch B, int64_t qi, int frame, int mode, float minX, float minY, float invW,
                                                 float invH, const uint8_t* taken, int lane)
{
    float x, y, r; int minLevel, maxLevel, cx0, cx1, cy0, cy1;
    proj_query(B, qi, mode, minX, minY, invW, invH, x, y, r, minLevel, maxLevel, cx0, cx1, cy0, cy1);      // (true: pass A got here)
    const int cap = B.cap;
    const pgorb_keypoint* K = B.K + (int64_t)frame * cap;
    const uint8_t* D = B.D + (int64_t)frame * cap * 32;
    const int32_t* gstart = B.gstart + (int64_t)frame * (GRID_CELLS + 1);
    const int32_t* gidx = B.gidx + (int64_t)frame * cap;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const uint4 q0 = reinterpret_cast<const uint4*>(B.desc + qi * 32)[0];
    const uint4 q1 = reinterpret_cast<const uint4*>(B.desc + qi * 32)[1];
    const float qangle = mode != 0 ? B.aux[qi] : 0.f;
    unsigned long long b1 = ~0ull, b2 = ~0ull;              // (distance << 48 | scan position << 32 | entry): the lane's two smallest
    int posBase = 0;
    for (int ix = cx0; ix <= cx1; ix++)
        for (int iy = cy0; iy <= cy1; iy++) {
            const int c = ix * GRID_ROWS + iy, s0 = gstart[c], cnt = gstart[c + 1] - s0;
            for (int k = lane; k < cnt; k += 64) {
                const int i2 = gidx[s0 + k];
                const pgorb_keypoint kp2 = K[i2];
                if (bCheckLevels && (kp2.octave < minLevel || (maxLevel >= 0 && kp2.octave > maxLevel))) continue;
                if (!(fabsf(__fsub_rn(kp2.x, x)) < r && fabsf(__fsub_rn(kp2.y, y)) < r)) continue;
                if (taken[i2]) continue;
                const int dist = sfi_distance(q0, q1, D + (int64_t)i2 * 32);
                const unsigned long long key = ((unsigned long long)dist << 48) | ((unsigned long long)(posBase + k) << 32) |
                                               proj_entry(dist, mode != 0 ? proj_bin(qangle, kp2.angle) : 0, kp2.octave, i2);
                if (key < b1) { b2 = b1; b1 = key; } else if (key < b2) b2 = key;
            }
            posBase += cnt;
        }
    // the wave's two smallest keys
    unsigned long long w1 = b1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(w1, d); w1 = o < w1 ? o : w1; }
    unsigned long long mine = (b1 == w1) ? b2 : b1, w2 = mine;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(w2, d); w2 = o < w2 ? o : w2; }
    return make_uint2(w1 == ~0ull ? 0xFFFFFFFFu : (uint32_t)w1, w2 == ~0ull ? 0xFFFFFFFFu : (uint32_t)w2);
}

// Pass B (round 4): the queries of a pair in the reference's ORDER without its sequence.  What query q decides depends on earlier
// queries only through the "holds a point" state of the keypoints in q's own list (:79-81 / :1397-1399 / :1542-1543), and a query
// only ever writes that state for the ONE keypoint it takes, a candidate of its list within the acceptance threshold (TH_HIGH, or
// ORBdist in the key-frame form): its "takeable" candidates.  So q can be decided as soon as no UNDECIDED earlier query has a takeable
// candidate in q's list -- and (mode 0 only: the second best of its ratio test reads candidates beyond TH_HIGH too; the best-only
// forms decide the same either way) q must not take a keypoint an undecided earlier query still has to read -- "deterministic reservations":
//   round:  minq[i]   = the smallest undecided query with keypoint i among its takeable candidates     (LDS atomicMin, all undecided in parallel)
//           minAny[i] = the smallest undecided query with keypoint i anywhere in its list              (mode 0)
//           q is ready  <=>  minq[i] >= q for every i in q's list (and minAny[i] >= q for every takeable i of it);
//           ready queries decide from the state as it is (reads only), then, behind a barrier, apply their decisions
//           (two ready queries never take the same keypoint: the later one would not be ready).
// The smallest undecided query is always ready, so the rounds end; map points project to different places, a query conflicts with a
// handful of neighbours, and about half of the undecided ones fall in every round.  One workgroup of 16 waves per pair, a wave per
// query and round, four queries' lists in flight per wave.  (Rounds 2-3: one wave walked the ~1 500-3 000 queries of a pair one after
// the other, ~0.4 us each: 1.64 ms for a single 3 200-point call against 0.58 ms on one CPU core; now 0.49 ms, and 127 pairs in
// 0.52 ms instead of 1.04.)  A query whose list did not fit the pool (LIST_OVER) waits until it is the smallest undecided one, holds
// back everything behind it, and is evaluated in place.  The rule against the plain sequence on random lists, without a GPU:
// tests/test_host_logic.py.  (SearchForInitialization keeps its sequential wave: see there.)
__global__ __launch_bounds__(RR_T) void k_search_by_projection(
    PgProjBatch B, float minX, float minY, float invW, float invH, int mode, float nnratio, int checkOrientation,
    PgLists Ls, int32_t* __restrict__ assignedOut, int32_t* __restrict__ nmatchesOut)
{
    const int p = blockIdx.x, frame = B.pairFrame ? B.pairFrame[p] : p;
    const int cap = B.cap, n = min(B.n[frame], cap), nq = min(B.nq[p], B.qcap);
    const uint8_t* kpHasPoint = B.kpHasPoint ? B.kpHasPoint + (int64_t)p * cap : nullptr;
    const int64_t qo = (int64_t)p * B.qcap;
    assignedOut += (int64_t)p * cap; nmatchesOut += p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int thTake = mode == 2 ? B.orbDist : TH_HIGH;
    // state in LDS: taken[i] = keypoint i holds a point (with observations, modes 0 / 1) before or by this call; asg[i] = query
    // assigned to keypoint i by this call; per query: list length, decision (keypoint, rotation bin), state
    uint32_t* minq = reinterpret_cast<uint32_t*>(pg_sfi_smem);            // [cap] smallest undecided query that may TAKE keypoint i
    uint32_t* minAny = minq + cap;                                        // [cap] smallest undecided query that LISTS keypoint i (mode 0: the second best of the ratio test)
    int32_t* asg = reinterpret_cast<int32_t*>(minAny + cap);               // [cap]
    int* ctrl = asg + cap;                                                // [8] counters, [8..40) the rotation histogram
    uint16_t* listA = reinterpret_cast<uint16_t*>(ctrl + 40);             // [qcap] undecided queries (two buffers)
    uint16_t* listB = listA + B.qcap;
    uint16_t* cntL = listB + B.qcap;                                      // [qcap]
    uint16_t* qBest = cntL + B.qcap;                                      // [qcap] keypoint the query takes
    int8_t* rotBin = reinterpret_cast<int8_t*>(qBest + B.qcap);           // [qcap] rotation bin of an accepted query (modes 1 / 2), or -1
    uint8_t* done = reinterpret_cast<uint8_t*>(rotBin + B.qcap);          // [qcap] 0 undecided, 1 decided to take qBest (to be applied), 2 finished
    uint8_t* taken = done + B.qcap;                                       // [cap]
    for (int i = tid; i < cap; i += RR_T) { taken[i] = (kpHasPoint && i < n) ? (kpHasPoint[i] != 0) : 0; asg[i] = -1; }
    if (tid < 40) ctrl[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nq; i += RR_T) {
        const uint16_t c = Ls.cnt[qo + i];
        cntL[i] = c; rotBin[i] = -1; done[i] = 0;
        if (c) listA[atomicAdd(&ctrl[0], 1)] = (uint16_t)i;
    }
    __syncthreads();
    uint16_t* cur = listA; uint16_t* nxt = listB;
    int curC = 0;
    while (true) {
        const int nun = ctrl[curC];
        if (nun == 0) break;
        for (int k = tid; k < cap; k += RR_T) { minq[k] = 0xFFFFFFFFu; minAny[k] = 0xFFFFFFFFu; }
        if (tid == 0) { ctrl[1 - curC] = 0; ctrl[3] = 0x7fffffff; ctrl[4] = 0x7fffffff; }
        __syncthreads();
        // ---- takeable candidates of every undecided query ----
        for (int u0 = wv * 4; u0 < nun; u0 += RR_W * 4) {
            int q[4], c[4]; uint32_t e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                q[j] = u0 + j < nun ? (int)cur[u0 + j] : -1;
                c[j] = q[j] >= 0 ? (int)cntL[q[j]] : 0;
                e[j] = (c[j] != (int)LIST_OVER && lane < min(c[j], LIST_K)) ? Ls.fixed[(qo + q[j]) * LIST_K + lane] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (q[j] < 0) break;
                if (lane == 0) atomicMin(&ctrl[4], q[j]);
                if (c[j] == (int)LIST_OVER) { if (lane == 0) atomicMin(&ctrl[3], q[j]); continue; }
                const uint32_t ovf = c[j] > LIST_K ? Ls.ovf[qo + q[j]] : 0u;
                for (int ch = 0; ch * 64 < c[j]; ch++) {
                    const uint32_t ee = ch == 0 ? e[j] : pg_list_chunk(Ls, qo + q[j], p, ovf, ch, lane);
                    if (ch * 64 + lane < c[j]) {
                        if (mode == 0) atomicMin(&minAny[ee & 0x3FFFu], (uint32_t)q[j]);
                        if ((int)(ee >> 23) <= thTake) atomicMin(&minq[ee & 0x3FFFu], (uint32_t)q[j]);
                    }
                }
            }
        }
        __syncthreads();
        const int minOver = ctrl[3], minAll = ctrl[4];
        // ---- ready queries decide ----
        for (int u0 = wv * 4; u0 < nun; u0 += RR_W * 4) {
            int q[4], c[4]; uint32_t e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                q[j] = u0 + j < nun ? (int)cur[u0 + j] : -1;
                c[j] = q[j] >= 0 ? (int)cntL[q[j]] : 0;
                e[j] = (c[j] != (int)LIST_OVER && lane < min(c[j], LIST_K)) ? Ls.fixed[(qo + q[j]) * LIST_K + lane] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int qq = q[j];
                if (qq < 0) break;
                const bool isOver = c[j] == (int)LIST_OVER;
                const uint32_t ovf = (!isOver && c[j] > LIST_K) ? Ls.ovf[qo + qq] : 0u;
                bool ready = isOver ? (qq == minAll) : (qq < minOver);
                if (ready && !isOver)
                    for (int ch = 0; ch * 64 < c[j]; ch++) {
                        const uint32_t ee = ch == 0 ? e[j] : pg_list_chunk(Ls, qo + qq, p, ovf, ch, lane);
                        // (mode 0 only: a query must not take a keypoint an undecided EARLIER query still has to read -- the second best of
                        //  its ratio test looks at candidates beyond TH_HIGH too; the best-only forms decide the same either way)
                        const bool blocked = minq[ee & 0x3FFFu] < (uint32_t)qq ||
                                             (mode == 0 && (int)(ee >> 23) <= thTake && minAny[ee & 0x3FFFu] < (uint32_t)qq);
                        if (__ballot(ch * 64 + lane < c[j] && blocked) != 0ull) { ready = false; break; }
                    }
                if (!ready) { if (lane == 0) nxt[atomicAdd(&ctrl[1 - curC], 1)] = (uint16_t)qq; continue; }
                // best and second best of the candidates that hold no point (:79-81 / :1397-1399 / :1542-1543); first minimum wins:
                // key = distance << 16 | position in the list
                uint32_t e1 = 0xFFFFFFFFu, e2 = 0xFFFFFFFFu;
                if (!isOver) {
                    unsigned w1 = 0xFFFFFFFFu, w2 = 0xFFFFFFFFu;
                    for (int ch = 0; ch * 64 < c[j]; ch++) {
                        const uint32_t ee = ch == 0 ? e[j] : pg_list_chunk(Ls, qo + qq, p, ovf, ch, lane);
                        const bool keep = ch * 64 + lane < c[j] && !taken[ee & 0x3FFFu];
                        const unsigned key = keep ? (((ee >> 23) << 16) | (unsigned)(ch * 64 + lane)) : 0xFFFFFFFFu;
                        unsigned k1, k2;
                        wave_min2_u32(key, k1, k2);
                        const uint32_t c1 = k1 == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readlane((int)ee, (int)(k1 & 63u));
                        const uint32_t c2 = k2 == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readlane((int)ee, (int)(k2 & 63u));
                        if (k1 < w1) {
                            if (w1 < k2) { w2 = w1; e2 = e1; } else { w2 = k2; e2 = c2; }
                            w1 = k1; e1 = c1;
                        } else if (k1 < w2) { w2 = k1; e2 = c1; }
                    }
                } else {
                    const uint2 ev = proj_eval_in_place(B, qo + qq, frame, mode, minX, minY, invW, invH, taken, lane);
                    e1 = ev.x; e2 = ev.y;
                }
                bool accept = false;
                int bin = -1, bestIdx = 0;
                if (e1 != 0xFFFFFFFFu && (int)(e1 >> 23) < 256) {                   // bestDist starts at 256 (:74 / :1390 / :1536)
                    const int bestDist = (int)(e1 >> 23);
                    bestIdx = (int)(e1 & 0x3FFFu);
                    if (mode == 0) {
                        const bool has2 = e2 != 0xFFFFFFFFu && (int)(e2 >> 23) < 256;
                        const int bestDist2 = has2 ? (int)(e2 >> 23) : 256;
                        const int bestLevel = (int)((e1 >> 14) & 15u), bestLevel2 = has2 ? (int)((e2 >> 14) & 15u) : -1;
                        if (bestDist <= TH_HIGH)                                    // :113-123
                            accept = !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2));
                    } else {
                        accept = bestDist <= thTake;                                // :1421 / :1554
                        if (accept && checkOrientation) bin = (int)((e1 >> 18) & 31u);     // :1426-1436 / :1559-1569 (computed in pass A)
                    }
                }
                if (lane == 0) {
                    if (accept) { qBest[qq] = (uint16_t)bestIdx; rotBin[qq] = (int8_t)bin; done[qq] = 1; }
                    else done[qq] = 2;
                }
            }
        }
        __syncthreads();
        // ---- apply: F.mvpMapPoints[bestIdx] = pMP (two queries of one round never take the same keypoint) ----
        for (int u = tid; u < nun; u += RR_T) {
            const int qq = cur[u];
            if (done[qq] != 1) continue;
            const int k = qBest[qq];
            asg[k] = qq;
            taken[k] = mode == 2 ? 1 : (B.hasObs[qo + qq] != 0);                  // (key-frame form: any point blocks, :1542-1543)
            atomicAdd(&ctrl[2], 1);
            done[qq] = 3;                                                           // accepted and applied
        }
        __syncthreads();
        uint16_t* t = cur; cur = nxt; nxt = t;
        curC = 1 - curC;
    }
    __syncthreads();
    if (mode != 0 && checkOrientation) {                       // :1443-1469 / :1575-1600
        int* hist = ctrl + 8;
        for (int i = tid; i < nq; i += RR_T) if (rotBin[i] >= 0) atomicAdd(&hist[rotBin[i]], 1);
        __syncthreads();
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        // rotHist[bin] holds bestIdx2 of every accepted query; every entry of a rejected bin resets
        // its keypoint to NULL and is counted out once (:1458-1465)
        int removed = 0;
        for (int i = tid; i < nq; i += RR_T) {
            const int bb = rotBin[i];
            if (bb >= 0 && bb != ind1 && bb != ind2 && bb != ind3) { asg[qBest[i]] = -1; removed++; }
        }
        if (removed) atomicSub(&ctrl[2], removed);
        __syncthreads();
    }
    for (int i = tid; i < cap; i += RR_T) assignedOut[i] = i < n ? asg[i] : -1;
    if (tid == 0) *nmatchesOut = ctrl[2];
}

// ---- SearchByBoW(KeyFrame*, Frame&), src/ORBmatcher.cc:161-290 -----------------------------------
// One wave per (key frame, frame) pair: merge-join of the two node lists; inside a common node the
// key frame's features are visited in order (each assignment removes a candidate for the later
// ones, :211-212) and the frame's features of that node are scanned one per lane.
// Batch layout (round 3): pair p = blockIdx.x, key frame pairKF[p] and frame pairF[p] of ONE extract batch (descriptors and
// keypoint angles `cap` apart); the FeatureVectors are the per-frame CSR arrays k_feature_vectors builds on the device
// (fvNode / fvFeat `cap` apart, fvStart cap + 1 apart); kfValid and the outputs are [npairs][cap].
struct PgBowBatch {
    const pgorb_keypoint* K; const uint8_t* D; const int32_t* n; int cap;
    const uint32_t* fvNode; const int32_t* fvStart; const uint32_t* fvFeat; const int32_t* nfv;
    const int32_t* pairKF; const int32_t* pairF; const uint8_t* kfValid;
};

// Round 3: NODES in parallel.  A frame feature belongs to exactly one vocabulary node, so the reference's order dependence
// ("vpMapPointMatches[realIdxF] already set", :219-220) never crosses a node: one wave walks ONE common node -- its key-frame
// features in order, the frame's features of the node one per lane and held in registers (descriptor, angle, "already
// matched" bit) for the whole walk -- and all nodes of all pairs run side by side.  A finishing wave per pair counts the
// matches and applies the rotation histogram (:256-277).  (The one-wave-per-pair form walked all ~2000 key-frame features of a
// pair in a row with the descriptor reads inside the chain: 1.34 ms per 127 pairs; a two-pass form like SearchByProjection's
// did not help because most features sit in nodes with more than 64 frame features.)
#define BOW_R 4                  // frame features per lane held in registers: nodes of up to 256 frame features
#define BOW_WAVES 64             // waves per pair, each takes the nodes a = wave, wave + 64, ...

__global__ __launch_bounds__(64) void k_search_by_bow(PgBowBatch B, float nnratio, int checkOrientation,
                                                       int32_t* __restrict__ matchesOut, int8_t* __restrict__ binOut)
{
    const int p = blockIdx.y, fa = B.pairKF[p], fb = B.pairF[p], cap = B.cap;
    const uint8_t* __restrict__ kfDesc = B.D + (int64_t)fa * cap * 32;
    const uint8_t* __restrict__ fDesc = B.D + (int64_t)fb * cap * 32;
    const pgorb_keypoint* __restrict__ kfK = B.K + (int64_t)fa * cap;
    const pgorb_keypoint* __restrict__ fK = B.K + (int64_t)fb * cap;
    const uint8_t* __restrict__ kfValid = B.kfValid + (int64_t)p * cap;
    const uint32_t* __restrict__ aNode = B.fvNode + (int64_t)fa * cap; const int32_t* __restrict__ aStart = B.fvStart + (int64_t)fa * (cap + 1);
    const uint32_t* __restrict__ aFeat = B.fvFeat + (int64_t)fa * cap;
    const uint32_t* __restrict__ bNode = B.fvNode + (int64_t)fb * cap; const int32_t* __restrict__ bStart = B.fvStart + (int64_t)fb * (cap + 1);
    const uint32_t* __restrict__ bFeat = B.fvFeat + (int64_t)fb * cap;
    const int nA = B.nfv[fa], nB = B.nfv[fb];
    matchesOut += (int64_t)p * cap; binOut += (int64_t)p * cap;
    const int lane = threadIdx.x;
    for (int a = blockIdx.x; a < nA; a += BOW_WAVES) {
        const uint32_t node = aNode[a];
        int lo = 0, hi = nB;                                              // the frame's entry of the same node (both lists ascend)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (bNode[mid] < node) lo = mid + 1; else hi = mid; }
        if (lo >= nB || bNode[lo] != node) continue;
        const int a0 = aStart[a], a1 = aStart[a + 1], b0 = bStart[lo], b1 = bStart[lo + 1], nb = b1 - b0;
        const bool inRegs = nb <= 64 * BOW_R;
        // the frame's features of the node: lane holds candidates k = lane, lane + 64, ... (k = position in the node's list)
        uint4 d0[BOW_R], d1[BOW_R]; float ang[BOW_R]; int idxF[BOW_R];
        unsigned takenBits = 0;
#pragma unroll
        for (int r = 0; r < BOW_R; r++) {
            const int k = 64 * r + lane;
            idxF[r] = -1; ang[r] = 0.f; d0[r] = make_uint4(0, 0, 0, 0); d1[r] = d0[r];
            if (inRegs && k < nb) {
                idxF[r] = (int)bFeat[b0 + k];
                d0[r] = reinterpret_cast<const uint4*>(fDesc + (int64_t)idxF[r] * 32)[0];
                d1[r] = reinterpret_cast<const uint4*>(fDesc + (int64_t)idxF[r] * 32)[1];
                ang[r] = fK[idxF[r]].angle;
            }
        }
        // The key frame's features of the node, one after the other.  Each needs its index (aFeat), then its descriptor and validity
        // through that index: two dependent global round trips, ~2 us per feature when they sat inside the iteration -- most of this
        // kernel's time.  They run two iterations / one iteration ahead instead (all lanes load the same addresses).
        int idxN = a0 < a1 ? (int)aFeat[a0] : 0, idxN2 = a0 + 1 < a1 ? (int)aFeat[a0 + 1] : 0;
        uint4 nq0 = make_uint4(0, 0, 0, 0), nq1 = nq0;
        uint8_t nvalid = 0;
        if (a0 < a1) {
            nq0 = reinterpret_cast<const uint4*>(kfDesc + (int64_t)idxN * 32)[0]; nq1 = reinterpret_cast<const uint4*>(kfDesc + (int64_t)idxN * 32)[1];
            nvalid = kfValid[idxN];
        }
        for (int ia = a0; ia < a1; ia++) {
            const int realIdxKF = __builtin_amdgcn_readfirstlane(idxN);
            const uint4 q0 = nq0, q1 = nq1;
            const bool valid = nvalid != 0;
            idxN = idxN2;
            idxN2 = ia + 2 < a1 ? (int)aFeat[ia + 2] : 0;
            if (ia + 1 < a1) {
                nq0 = reinterpret_cast<const uint4*>(kfDesc + (int64_t)idxN * 32)[0]; nq1 = reinterpret_cast<const uint4*>(kfDesc + (int64_t)idxN * 32)[1];
                nvalid = kfValid[idxN];
            }
            if (!valid) continue;                                         // !pMP || pMP->isBad() (:208-213)
            unsigned b1key = 0xFFFFFFFFu, b2key = 0xFFFFFFFFu;
            if (inRegs) {
#pragma unroll
                for (int r = 0; r < BOW_R; r++) {
                    if (idxF[r] < 0 || (takenBits >> r) & 1u) continue;   // vpMapPointMatches[realIdxF] (:219-220)
                    const int dist = __popc(q0.x ^ d0[r].x) + __popc(q0.y ^ d0[r].y) + __popc(q0.z ^ d0[r].z) + __popc(q0.w ^ d0[r].w) +
                                     __popc(q1.x ^ d1[r].x) + __popc(q1.y ^ d1[r].y) + __popc(q1.z ^ d1[r].z) + __popc(q1.w ^ d1[r].w);
                    const unsigned key = ((unsigned)dist << 16) | (unsigned)(64 * r + lane);
                    if (key < b1key) { b2key = b1key; b1key = key; } else if (key < b2key) b2key = key;
                }
            } else {                                                      // a node with more frame features than the registers hold
                for (int k = lane; k < nb; k += 64) {
                    const int realIdxF = (int)bFeat[b0 + k];
                    if (matchesOut[realIdxF] >= 0) continue;              // (this wave's own earlier writes: same lane, program order)
                    const unsigned key = ((unsigned)sfi_distance(q0, q1, fDesc + (int64_t)realIdxF * 32) << 16) | (unsigned)k;
                    if (key < b1key) { b2key = b1key; b1key = key; } else if (key < b2key) b2key = key;
                }
            }
            const unsigned w1 = wave_min_u32(b1key);
            if (w1 == 0xFFFFFFFFu || (int)(w1 >> 16) >= 256) continue;     // bestDist1 starts at 256
            const unsigned w2 = wave_min_u32(b1key == w1 ? b2key : b1key);
            const int bestDist1 = (int)(w1 >> 16);
            const int bestDist2 = (w2 != 0xFFFFFFFFu && (int)(w2 >> 16) < 256) ? (int)(w2 >> 16) : 256;
            if (bestDist1 <= TH_LOW && (float)bestDist1 < __fmul_rn(nnratio, (float)bestDist2)) {   // :233-235
                const int kbest = (int)(w1 & 0xFFFFu);
                if ((kbest & 63) == lane) {                              // the lane that holds the winner files it
                    int bestIdxF; float fang;
                    if (inRegs) {
                        const int r = kbest >> 6;
                        bestIdxF = idxF[0]; fang = ang[0];
#pragma unroll
                        for (int rr = 1; rr < BOW_R; rr++) if (r == rr) { bestIdxF = idxF[rr]; fang = ang[rr]; }
                        takenBits |= 1u << r;
                    } else {
                        bestIdxF = (int)bFeat[b0 + kbest]; fang = fK[bestIdxF].angle;
                    }
                    matchesOut[bestIdxF] = realIdxKF;
                    binOut[bestIdxF] = (int8_t)(checkOrientation ? proj_bin(kfK[realIdxKF].angle, fang) : -1);    // :241-250
                }
                if (!inRegs) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the next feature's scan reads matchesOut from other lanes
            }
        }
    }
}

// after the nodes: count the matches of a pair and apply the rotation histogram
__global__ __launch_bounds__(64) void k_bow_finish(PgBowBatch B, int checkOrientation, int32_t* __restrict__ matchesOut,
                                                    const int8_t* __restrict__ binIn, int32_t* __restrict__ nmatchesOut)
{
    const int p = blockIdx.x, fb = B.pairF[p], cap = B.cap, nf = min(B.n[fb], cap), lane = threadIdx.x;
    matchesOut += (int64_t)p * cap; binIn += (int64_t)p * cap; nmatchesOut += p;
    int nmatches = 0;
    for (int i = lane; i < nf; i += 64) nmatches += matchesOut[i] >= 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmatches += __shfl_xor(nmatches, d);
    const int8_t* rotBin = binIn;
    int32_t* asg = matchesOut;
    if (checkOrientation) {                                               // :256-277
        // the 30 bin sizes: lanes stride over the features, one LDS atomic each (every lane walking all nf bins by itself, a dependent
        // byte load per feature, was 90 of this kernel's 93 us at 2 000 features)
        __shared__ int hist[64];
        hist[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < nf; i += 64) { const int bb = rotBin[i]; if (bb >= 0) atomicAdd(&hist[bb & 63], 1); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int h = hist[lane];
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sc = __shfl(h, i);
            if (sc > max1) { max3 = max2; max2 = max1; max1 = sc; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sc > max2) { max3 = max2; max2 = sc; ind3 = ind2; ind2 = i; }
            else if (sc > max3) { max3 = sc; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        int removed = 0;
        for (int i = lane; i < nf; i += 64) {
            const int bb = rotBin[i];
            if (bb >= 0 && bb != ind1 && bb != ind2 && bb != ind3 && asg[i] >= 0) { asg[i] = -1; removed++; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) removed += __shfl_xor(removed, d);
        nmatches -= removed;
    }
    if (lane == 0) *nmatchesOut = nmatches;
}

// FeatureVector of every frame of a batch (DBoW2 FeatureVector::addFeature, FeatureVector.cpp:31-45, as
// TemplatedVocabulary::transform fills it, TemplatedVocabulary.h:1180-1186): map<node id, vector<feature index>> with the
// indices appended in feature order = the features sorted by (node id, index), as CSR.  One workgroup per frame: rank of
// every feature by counting (n <= a few thousand: n^2 / 1024 compares per thread on LDS), scatter, group heads by a scan.
__global__ __launch_bounds__(1024) void k_feature_vectors(const uint32_t* __restrict__ node, const int32_t* __restrict__ nIn, int cap,
                                                          uint32_t* __restrict__ fvNode, int32_t* __restrict__ fvStart,
                                                          uint32_t* __restrict__ fvFeat, int32_t* __restrict__ nfv)
{
    const int f = blockIdx.x, tid = threadIdx.x, n = min(nIn[f], cap);
    uint32_t* key = reinterpret_cast<uint32_t*>(pg_sfi_smem);            // [cap] node id of feature i
    uint32_t* snode = key + cap;                                          // [cap] sorted node ids
    int* scan = reinterpret_cast<int*>(snode + cap);                      // [1024 + 1]
    node += (int64_t)f * cap; fvNode += (int64_t)f * cap; fvFeat += (int64_t)f * cap; fvStart += (int64_t)f * (cap + 1);
    for (int i = tid; i < n; i += 1024) key[i] = node[i];
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const uint32_t k = key[i];
        int r = 0;
        for (int j = 0; j < n; j++) { const uint32_t kj = key[j]; r += (kj < k) || (kj == k && j < i); }
        snode[r] = k; fvFeat[r] = (uint32_t)i;
    }
    __syncthreads();
    // group heads: position r starts a group when its node differs from its predecessor's; exclusive scan of the flags
    const int per = (n + 1023) / 1024, r0 = tid * per, r1 = min(n, r0 + per);
    int heads = 0;
    for (int r = r0; r < r1; r++) heads += (r == 0 || snode[r] != snode[r - 1]);
    scan[tid] = heads;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int t = 0; t < 1024; t++) { const int h = scan[t]; scan[t] = acc; acc += h; } scan[1024] = acc; }
    __syncthreads();
    int g = scan[tid];
    for (int r = r0; r < r1; r++)
        if (r == 0 || snode[r] != snode[r - 1]) { fvNode[g] = snode[r]; fvStart[g] = r; g++; }
    if (tid == 0) { fvStart[scan[1024]] = n; nfv[f] = scan[1024]; }
}

// The same CSR by SORTING (round 4; round 5: the sort is this file's own): the features' keys node id << 13 | feature index are
// unique, the features arrive in index order, so the FeatureVector is a STABLE sort by node id.  One workgroup per frame holds up to
// FV_T * FV_IPT = 8 192 keys in LDS and runs least-significant-digit passes of 2 bits over exactly the bits the largest node id uses
// (ORBvoc at levelsup 4: 11 bits, six passes): a thread owns 8 CONSECUTIVE positions (stability), counts its four digits in two
// packed 16 + 16-bit words (a count never exceeds 8 192), one DPP wave scan per word + sixteen wave totals give every thread the
// number of equal digits in front of it, and the keys are scattered into the second buffer.  (Round 4 called rocPRIM's
// block_radix_sort here; the counting form above is O(n^2) -- 0.17 ms for 128 frames of 2 000 features, 0.65 ms at 4 000 -- and stays
// for frames beyond 8 192 features.)
#define FV_T 1024
#define FV_IPT 8
__global__ __launch_bounds__(FV_T) void k_feature_vectors_sorted(const uint32_t* __restrict__ node, const int32_t* __restrict__ nIn, int cap,
                                                                 uint32_t* __restrict__ fvNode, int32_t* __restrict__ fvStart,
                                                                 uint32_t* __restrict__ fvFeat, int32_t* __restrict__ nfv)
{
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = min(nIn[f], cap);
    unsigned long long* bufA = reinterpret_cast<unsigned long long*>(pg_sfi_smem);                      // [FV_T * FV_IPT] keys
    unsigned long long* bufB = bufA + FV_T * FV_IPT;
    uint32_t* snode = reinterpret_cast<uint32_t*>(bufB);                                                  // the sorted node ids end up here
    int* wsum = reinterpret_cast<int*>(bufB + FV_T * FV_IPT);                                             // [2 * FV_T / 64 + 2]
    node += (int64_t)f * cap; fvNode += (int64_t)f * cap; fvFeat += (int64_t)f * cap; fvStart += (int64_t)f * (cap + 1);
    uint32_t mx = 0;
#pragma unroll
    for (int k = 0; k < FV_IPT; k++) {
        const int i = tid * FV_IPT + k;
        const uint32_t nd = i < n ? node[i] : 0u;
        mx = max(mx, nd);
        bufA[i] = i < n ? (((unsigned long long)nd << 13) | (unsigned long long)i) : 0xFFFFFFFFFFFFFFFFull;      // padding: all ones, stays last
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
    if (lane == 0) wsum[wv] = (int)mx;
    __syncthreads();
    for (int w = 0; w < FV_T / 64; w++) mx = max(mx, (uint32_t)wsum[w]);
    __syncthreads();
    const int nbits = 32 - __clz(mx | 1u);                      // bits of the largest node id
    unsigned long long *src = bufA, *dst = bufB;
    for (int shift = 13; shift < 13 + nbits; shift += 2) {
        unsigned long long key[FV_IPT];
        uint32_t c01 = 0u, c23 = 0u, before[FV_IPT];            // packed digit counts of this thread: (digit 0 | digit 1 << 16), (2 | 3 << 16)
#pragma unroll
        for (int k = 0; k < FV_IPT; k++) {
            key[k] = src[tid * FV_IPT + k];
            const uint32_t d = (uint32_t)(key[k] >> shift) & 3u, fld = (d & 1u) << 4;
            const uint32_t word = (d & 2u) ? c23 : c01;
            before[k] = (word >> fld) & 0xFFFFu;                // equal digits of this thread in front of key k
            if (d & 2u) c23 += 1u << fld; else c01 += 1u << fld;
        }
        const uint32_t i01 = (uint32_t)wave_incl_scan((int)c01, lane), i23 = (uint32_t)wave_incl_scan((int)c23, lane);
        if (lane == 63) { wsum[2 * wv] = (int)i01; wsum[2 * wv + 1] = (int)i23; }
        __syncthreads();
        uint32_t b01 = 0u, b23 = 0u, t01 = 0u, t23 = 0u;        // digits in the waves in front of this one / in the whole block
        for (int w = 0; w < FV_T / 64; w++) {
            const uint32_t v01 = (uint32_t)wsum[2 * w], v23 = (uint32_t)wsum[2 * w + 1];
            if (w < wv) { b01 += v01; b23 += v23; }
            t01 += v01; t23 += v23;
        }
        const uint32_t e01 = b01 + i01 - c01, e23 = b23 + i23 - c23;             // exclusive over the threads, still packed
        const uint32_t base1 = t01 & 0xFFFFu, base2 = base1 + (t01 >> 16), base3 = base2 + (t23 & 0xFFFFu);
#pragma unroll
        for (int k = 0; k < FV_IPT; k++) {
            const uint32_t d = (uint32_t)(key[k] >> shift) & 3u, fld = (d & 1u) << 4;
            const uint32_t ex = (((d & 2u) ? e23 : e01) >> fld) & 0xFFFFu;
            const uint32_t base = d == 0u ? 0u : d == 1u ? base1 : d == 2u ? base2 : base3;
            dst[base + ex + before[k]] = key[k];
        }
        __syncthreads();
        unsigned long long* t = src; src = dst; dst = t;
    }
    unsigned long long skey[FV_IPT];
#pragma unroll
    for (int k = 0; k < FV_IPT; k++) skey[k] = src[tid * FV_IPT + k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FV_IPT; k++) {
        const int r = tid * FV_IPT + k;
        if (r < n) { snode[r] = (uint32_t)(skey[k] >> 13); fvFeat[r] = (uint32_t)(skey[k] & 8191ull); }
    }
    __syncthreads();
    // group heads: position r starts a group when its node differs from its predecessor's; exclusive scan of the counts over the threads
    int heads = 0;
#pragma unroll
    for (int k = 0; k < FV_IPT; k++) { const int r = tid * FV_IPT + k; heads += (r < n && (r == 0 || snode[r] != snode[r - 1])) ? 1 : 0; }
    const int incl = wave_incl_scan(heads, lane);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < FV_T / 64; w++) { const int v = wsum[w]; base += w < wv ? v : 0; total += v; }
    int g = base + incl - heads;
#pragma unroll
    for (int k = 0; k < FV_IPT; k++) {
        const int r = tid * FV_IPT + k;
        if (r < n && (r == 0 || snode[r] != snode[r - 1])) { fvNode[g] = snode[r]; fvStart[g] = r; g++; }
    }
    if (tid == 0) { fvStart[total] = n; nfv[f] = total; }
}

// ---- cv::undistortPoints (OpenCV 2.4 imgproc/undistort.cpp cvUndistortPoints), 5 fixed-point
// iterations in double, R = identity, P = K.  Same operation order on host and device, no FMA.
struct PgCamera { double fx, fy, cx, cy, ifx, ify, k1, k2, p1, p2, k3; };

__host__ __device__ inline void pg_undistort_point(const PgCamera& C, float px, float py, float* ox, float* oy)
{
#ifdef __HIP_DEVICE_COMPILE__
#define PGM(a, b) __dmul_rn((a), (b))
#define PGA(a, b) __dadd_rn((a), (b))
#define PGS(a, b) __dsub_rn((a), (b))
#define PGD(a, b) __ddiv_rn((a), (b))
#else
#define PGM(a, b) ((a) * (b))
#define PGA(a, b) ((a) + (b))
#define PGS(a, b) ((a) - (b))
#define PGD(a, b) ((a) / (b))
#endif
    double x = (double)px, y = (double)py;
    const double x0 = x = PGM(PGS(x, C.cx), C.ifx);
    const double y0 = y = PGM(PGS(y, C.cy), C.ify);
    for (int j = 0; j < 5; j++) {
        const double r2 = PGA(PGM(x, x), PGM(y, y));
        // icdist = (1 + ((k[7]*r2 + k[6])*r2 + k[5])*r2) / (1 + ((k[4]*r2 + k[1])*r2 + k[0])*r2), k5..k7 = 0
        const double icdist = PGD(1.0, PGA(1.0, PGM(PGA(PGM(PGA(PGM(C.k3, r2), C.k2), r2), C.k1), r2)));
        const double deltaX = PGA(PGM(PGM(PGM(2.0, C.p1), x), y), PGM(C.p2, PGA(r2, PGM(PGM(2.0, x), x))));
        const double deltaY = PGA(PGM(C.p1, PGA(r2, PGM(PGM(2.0, y), y))), PGM(PGM(PGM(2.0, C.p2), x), y));
        x = PGM(PGS(x0, deltaX), icdist);
        y = PGM(PGS(y0, deltaY), icdist);
    }
    // RR = P * R = K: xx = fx*x + 0*y + cx, ww = 1/(0*x + 0*y + 1)
    const double xx = PGA(PGA(PGM(C.fx, x), PGM(0.0, y)), C.cx);
    const double yy = PGA(PGA(PGM(0.0, x), PGM(C.fy, y)), C.cy);
    const double ww = PGD(1.0, PGA(PGA(PGM(0.0, x), PGM(0.0, y)), 1.0));
    *ox = (float)PGM(xx, ww);
    *oy = (float)PGM(yy, ww);
#undef PGM
#undef PGA
#undef PGS
#undef PGD
}

static PgCamera pg_make_camera(const float camera[4], const float dist[5])
{
    PgCamera C;
    C.fx = camera[0]; C.fy = camera[1]; C.cx = camera[2]; C.cy = camera[3];
    C.ifx = 1. / C.fx; C.ify = 1. / C.fy;
    C.k1 = dist[0]; C.k2 = dist[1]; C.p1 = dist[2]; C.p2 = dist[3]; C.k3 = dist[4];
    return C;
}

__global__ __launch_bounds__(256) void k_undistort_keypoints(const pgorb_keypoint* __restrict__ in,
                                                              const int32_t* __restrict__ nper, int cap,
                                                              PgCamera C, int identity, pgorb_keypoint* __restrict__ out)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(nper[f], cap)) return;
    pgorb_keypoint k = in[(int64_t)f * cap + i];
    if (!identity) pg_undistort_point(C, k.x, k.y, &k.x, &k.y);
    out[(int64_t)f * cap + i] = k;
}

extern "C" {

int pgorb_undistort_keypoints_batch_device(pgorb_ctx* c, const pgorb_keypoint* d_kps, const int32_t* d_n, int nframes,
                                           int cap, const float camera[4], const float dist[5], pgorb_keypoint* d_out,
                                           void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!d_kps || !d_n || !d_out || nframes < 1 || cap < 1 || !camera || !dist)
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_undistort_keypoints_batch_device");
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    hipLaunchKernelGGL(k_undistort_keypoints, dim3((cap + 255) / 256, nframes), dim3(256), 0, (hipStream_t)stream, d_kps, d_n,
                       cap, pg_make_camera(camera, dist), dist[0] == 0.0f ? 1 : 0, d_out);      // Frame.cc:410
    if (hipGetLastError() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_undistort_keypoints launch failed");
    return 0;
}

int pgorb_undistort_keypoints(pgorb_ctx* c, const pgorb_keypoint* kps, int n, const float camera[4], const float dist[5],
                              pgorb_keypoint* out)
{
    if (!c) return PGORB_E_ARG;
    if (n < 0 || (n && (!kps || !out)) || !camera || !dist) return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_undistort_keypoints");
    if (!n) return 0;
    void* dv;
    int rc = pg_ctx_stage(c, 0, (size_t)2 * n * sizeof(pgorb_keypoint) + 128, &dv);
    if (rc) return rc;
    pgorb_keypoint* din = (pgorb_keypoint*)dv;
    pgorb_keypoint* dout = din + n;
    int32_t* dn = (int32_t*)(((uintptr_t)(dout + n) + 63) & ~(uintptr_t)63);
    if (hipMemcpy(din, kps, (size_t)n * sizeof(pgorb_keypoint), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dn, &n, 4, hipMemcpyHostToDevice) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy H2D failed");
    if ((rc = pgorb_undistort_keypoints_batch_device(c, din, dn, 1, n, camera, dist, dout, 0))) return rc;
    if (hipMemcpy(out, dout, (size_t)n * sizeof(pgorb_keypoint), hipMemcpyDeviceToHost) != hipSuccess)
        return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy D2H failed");
    return 0;
}

int pgorb_image_bounds(int cols, int rows, const float camera[4], const float dist[5], float bounds[4])
{
    if (cols < 1 || rows < 1 || !camera || !dist || !bounds) return PGORB_E_ARG;
    if (dist[0] == 0.0f) {                                     // Frame.cc:461-466
        bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
        return 0;
    }
    const PgCamera C = pg_make_camera(camera, dist);
    const float cx[4] = {0.0f, (float)cols, 0.0f, (float)cols}, cy[4] = {0.0f, 0.0f, (float)rows, (float)rows};
    float ux[4], uy[4];
    for (int i = 0; i < 4; i++) pg_undistort_point(C, cx[i], cy[i], &ux[i], &uy[i]);
    bounds[0] = std::min(ux[0], ux[2]); bounds[1] = std::max(ux[1], ux[3]);     // Frame.cc:455-458
    bounds[2] = std::min(uy[0], uy[1]); bounds[3] = std::max(uy[2], uy[3]);
    return 0;
}

// per-device "dynamic LDS limit already raised to" bookkeeping of the three latency kernels below
static bool pg_raise_lds(pgorb_ctx* c, const void* fn, int which, size_t lds)
{
    static size_t configured[7][64] = {{0}};
    const int dv = pg_ctx_device(c) & 63;
    if (lds > 160 * 1024) return false;
    if (lds > configured[which][dv]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        configured[which][dv] = lds;
    }
    return true;
}

int pgorb_feature_vectors_batch_device(pgorb_ctx* c, const uint32_t* d_node, const int32_t* d_n, int nframes, int cap,
                                       uint32_t* d_fv_node, int32_t* d_fv_start, uint32_t* d_fv_feat, int32_t* d_nfv, void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!d_node || !d_n || nframes < 1 || cap < 1 || !d_fv_node || !d_fv_start || !d_fv_feat || !d_nfv)
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_feature_vectors_batch_device");
    if (cap > 16000) return pg_ctx_fail(c, PGORB_E_LIMIT, "more than 16000 keypoints per frame");
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    static const bool counting = getenv("PGORB_FV_COUNTING") != nullptr;      // (A / B switch: the O(n^2) counting form for every size)
    if (cap <= FV_T * FV_IPT && !counting) {
        const size_t ldsS = (size_t)2 * FV_T * FV_IPT * 8 + (2 * FV_T / 64 + 2) * 4;       // two key buffers + the wave totals
        if (!pg_raise_lds(c, reinterpret_cast<const void*>(k_feature_vectors_sorted), 6, ldsS)) return pg_ctx_fail(c, PGORB_E_LIMIT, "feature vector scratch exceeds the LDS");
        hipLaunchKernelGGL(k_feature_vectors_sorted, dim3(nframes), dim3(FV_T), ldsS, (hipStream_t)stream, d_node, d_n, cap, d_fv_node, d_fv_start, d_fv_feat, d_nfv);
        if (hipGetLastError() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_feature_vectors_sorted launch failed");
        return 0;
    }
    const size_t lds = (size_t)cap * 8 + 1025 * 4;
    if (!pg_raise_lds(c, reinterpret_cast<const void*>(k_feature_vectors), 2, lds)) return pg_ctx_fail(c, PGORB_E_LIMIT, "feature vector scratch exceeds the LDS");
    hipLaunchKernelGGL(k_feature_vectors, dim3(nframes), dim3(1024), lds, (hipStream_t)stream, d_node, d_n, cap, d_fv_node, d_fv_start, d_fv_feat, d_nfv);
    if (hipGetLastError() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_feature_vectors launch failed");
    return 0;
}

int pgorb_search_by_bow_batch_device(pgorb_ctx* c, const pgorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int cap,
                                     const uint32_t* d_fv_node, const int32_t* d_fv_start, const uint32_t* d_fv_feat, const int32_t* d_nfv,
                                     const int32_t* d_pair_kf, const int32_t* d_pair_f, int npairs, const uint8_t* d_kf_point_valid,
                                     float nnratio, int check_orientation, int32_t* d_matches, int32_t* d_nmatches, void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!d_kps || !d_desc || !d_n || cap < 1 || !d_fv_node || !d_fv_start || !d_fv_feat || !d_nfv || npairs < 0 ||
        (npairs && (!d_pair_kf || !d_pair_f || !d_kf_point_valid || !d_matches || !d_nmatches)))
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_search_by_bow_batch_device");
    if (cap > 16000) return pg_ctx_fail(c, PGORB_E_LIMIT, "more than 16000 keypoints per frame");
    if (!npairs) return 0;
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    // scratch: the rotation bin of every matched frame feature [npairs][cap] i8
    void* scratch;
    int rcs = pg_ctx_scratch(c, (size_t)npairs * cap + 256, (hipStream_t)stream, &scratch);
    if (rcs) return rcs;
    int8_t* bins = (int8_t*)scratch;
    if (hipMemsetAsync(d_matches, 0xFF, (size_t)npairs * cap * 4, (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(bins, 0xFF, (size_t)npairs * cap, (hipStream_t)stream) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipMemsetAsync failed");
    PgBowBatch B = {d_kps, d_desc, d_n, cap, d_fv_node, d_fv_start, d_fv_feat, d_nfv, d_pair_kf, d_pair_f, d_kf_point_valid};
    hipLaunchKernelGGL(k_search_by_bow, dim3(BOW_WAVES, npairs), dim3(64), 0, (hipStream_t)stream, B, nnratio, check_orientation, d_matches, bins);
    hipLaunchKernelGGL(k_bow_finish, dim3(npairs), dim3(64), 0, (hipStream_t)stream, B, check_orientation, d_matches, bins, d_nmatches);
    if (hipGetLastError() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_search_by_bow launch failed");
    return pg_ctx_scratch_done(c, (hipStream_t)stream);
}

// single pair through host buffers: the pair becomes a two-frame batch (key frame = frame 0, frame = frame 1)
int pgorb_search_by_bow(pgorb_ctx* c, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_point_valid, int nkf,
                        const uint32_t* kf_fv_node, const int32_t* kf_fv_start, const uint32_t* kf_fv_feat, int kf_nfv,
                        const uint8_t* f_desc, const float* f_angle, int nf, const uint32_t* f_fv_node,
                        const int32_t* f_fv_start, const uint32_t* f_fv_feat, int f_nfv, float nnratio,
                        int check_orientation, int32_t* matches)
{
    if (!c) return PGORB_E_ARG;
    if (nkf < 0 || nf < 0 || kf_nfv < 0 || f_nfv < 0 || (nf && !matches) ||
        (nkf && (!kf_desc || !kf_angle || !kf_point_valid)) || (nf && (!f_desc || !f_angle)) ||
        (kf_nfv && (!kf_fv_node || !kf_fv_start || !kf_fv_feat)) || (f_nfv && (!f_fv_node || !f_fv_start || !f_fv_feat)))
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_search_by_bow");
    for (int i = 0; i < nf; i++) matches[i] = -1;
    if (!nkf || !nf || !kf_nfv || !f_nfv) return 0;
    if (nf > 16000 || nkf > 16000) return pg_ctx_fail(c, PGORB_E_LIMIT, "more than 16000 keypoints");
    const int cap = std::max(nkf, nf);
    if (kf_nfv > cap || f_nfv > cap || kf_fv_start[kf_nfv] > nkf || f_fv_start[f_nfv] > nf)
        return pg_ctx_fail(c, PGORB_E_ARG, "pgorb_search_by_bow: FeatureVector names more features than the frame has");
    auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t oK = place((size_t)2 * cap * sizeof(pgorb_keypoint)), oD = place((size_t)2 * cap * 32), oN = place(8), oV = place(cap),
                 oFN = place((size_t)2 * cap * 4), oFS = place((size_t)2 * (cap + 1) * 4), oFF = place((size_t)2 * cap * 4), oNF = place(8),
                 oP = place(8), oM = place((size_t)cap * 4), oNM = place(64);
    void *dv, *hv;
    int rc = pg_ctx_stage(c, 0, off, &dv);
    if (rc) return rc;
    // everything the host supplies sits before oM: one upload from the page-locked buffer; oM.. is one download
    const size_t upBytes = oM, downBytes = off - oM;
    if ((rc = pg_ctx_pinned(c, std::max(upBytes, downBytes), &hv))) return rc;
    uint8_t* d = (uint8_t*)dv; uint8_t* h = (uint8_t*)hv;
    pgorb_keypoint* kk = (pgorb_keypoint*)(h + oK);
    memset(kk, 0, (size_t)2 * cap * sizeof(pgorb_keypoint));
    for (int i = 0; i < nkf; i++) kk[i].angle = kf_angle[i];
    for (int i = 0; i < nf; i++) kk[(size_t)cap + i].angle = f_angle[i];
    const int32_t nn[2] = {nkf, nf}, nfvs[2] = {kf_nfv, f_nfv}, pr[2] = {0, 1};
    auto up = [&](size_t o, const void* p, size_t n) { if (n) memcpy(h + o, p, n); };
    up(oD, kf_desc, (size_t)nkf * 32); up(oD + (size_t)cap * 32, f_desc, (size_t)nf * 32);
    up(oN, nn, 8); up(oV, kf_point_valid, nkf);
    up(oFN, kf_fv_node, (size_t)kf_nfv * 4); up(oFN + (size_t)cap * 4, f_fv_node, (size_t)f_nfv * 4);
    up(oFS, kf_fv_start, (size_t)(kf_nfv + 1) * 4); up(oFS + (size_t)(cap + 1) * 4, f_fv_start, (size_t)(f_nfv + 1) * 4);
    up(oFF, kf_fv_feat, (size_t)kf_fv_start[kf_nfv] * 4); up(oFF + (size_t)cap * 4, f_fv_feat, (size_t)f_fv_start[f_nfv] * 4);
    up(oNF, nfvs, 8); up(oP, pr, 8);
    if (hipMemcpyAsync(d, h, upBytes, hipMemcpyHostToDevice, 0) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy H2D failed");
    rc = pgorb_search_by_bow_batch_device(c, (const pgorb_keypoint*)(d + oK), d + oD, (const int32_t*)(d + oN), cap, (const uint32_t*)(d + oFN),
                                          (const int32_t*)(d + oFS), (const uint32_t*)(d + oFF), (const int32_t*)(d + oNF), (const int32_t*)(d + oP),
                                          (const int32_t*)(d + oP) + 1, 1, d + oV, nnratio, check_orientation, (int32_t*)(d + oM), (int32_t*)(d + oNM), nullptr);
    if (rc) return rc;
    if (hipMemcpyAsync(h, d + oM, downBytes, hipMemcpyDeviceToHost, 0) != hipSuccess || hipStreamSynchronize(0) != hipSuccess)
        return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy D2H failed");
    memcpy(matches, h, (size_t)nf * 4);
    int32_t nm;
    memcpy(&nm, h + (oNM - oM), 4);
    return nm;
}

int pgorb_frame_grid_batch_device(pgorb_ctx* c, const pgorb_keypoint* d_kps, const int32_t* d_n, int nframes,
                                  int cap, float min_x, float max_x, float min_y, float max_y,
                                  int32_t* d_grid_start, int32_t* d_grid_idx, void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!d_kps || !d_n || nframes < 1 || cap < 1 || !d_grid_start || !d_grid_idx || !(max_x > min_x) || !(max_y > min_y))
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_frame_grid_batch_device");
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    const float invW = (float)GRID_COLS / (max_x - min_x), invH = (float)GRID_ROWS / (max_y - min_y);   // Frame.cc:216-217
    hipLaunchKernelGGL(k_frame_grid, dim3(nframes), dim3(256), 0, (hipStream_t)stream, d_kps, d_n, cap, min_x, min_y,
                       invW, invH, d_grid_start, d_grid_idx);
    if (hipGetLastError() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_frame_grid launch failed");
    return 0;
}

int pgorb_search_for_initialization_batch_device(pgorb_ctx* c, const pgorb_keypoint* d_kps, const uint8_t* d_desc,
                                    const int32_t* d_n, int cap, const int32_t* d_grid_start,
                                    const int32_t* d_grid_idx, const int32_t* d_pair_f1, const int32_t* d_pair_f2,
                                    int npairs, float min_x, float max_x, float min_y, float max_y,
                                    float* d_prev_matched, int32_t* d_matches12, int32_t* d_nmatches,
                                    int window_size, float nnratio, int check_orientation, void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!d_kps || !d_desc || !d_n || cap < 1 || !d_grid_start || !d_grid_idx || npairs < 0 ||
        (npairs && (!d_pair_f1 || !d_pair_f2 || !d_prev_matched || !d_matches12 || !d_nmatches)) ||
        !(max_x > min_x) || !(max_y > min_y) || window_size < 0)
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_search_for_initialization_batch_device");
    if (cap > 16000) return pg_ctx_fail(c, PGORB_E_LIMIT, "more than 16000 keypoints per frame");
    if (!npairs) return 0;
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    const float invW = (float)GRID_COLS / (max_x - min_x), invH = (float)GRID_ROWS / (max_y - min_y);
    // scratch of the two passes: the candidate lists of every F1 keypoint of every pair (PgLists)
    PgLists Ls;
    const size_t szL = pg_lists_layout(nullptr, npairs, cap, &Ls);
    void* scratch;
    int rc = pg_ctx_scratch(c, szL + 256, (hipStream_t)stream, &scratch);
    if (rc) return rc;
    pg_lists_layout(scratch, npairs, cap, &Ls);
    const size_t ldsA = (size_t)4 * cap * 2, ldsB = (size_t)cap * 10 + 192;      // (+ the 32-bin histogram)
    if (!pg_raise_lds(c, reinterpret_cast<const void*>(k_sfi_candidates), 4, ldsA) ||
        !pg_raise_lds(c, reinterpret_cast<const void*>(k_search_for_initialization), 5, ldsB))
        return pg_ctx_fail(c, PGORB_E_LIMIT, "SearchForInitialization state exceeds the LDS");
    if (hipMemsetAsync(Ls.poolTop, 0, (size_t)npairs * 4, (hipStream_t)stream) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipMemsetAsync failed");
    hipLaunchKernelGGL(k_sfi_candidates, dim3((cap + 3) / 4, npairs), dim3(256), ldsA, (hipStream_t)stream, d_kps, d_desc, d_n, cap,
                       d_grid_start, d_grid_idx, d_pair_f1, d_pair_f2, min_x, min_y, invW, invH, d_prev_matched, window_size, Ls);
    hipLaunchKernelGGL(k_search_for_initialization, dim3(npairs), dim3(64), ldsB, (hipStream_t)stream, d_kps, d_desc,
                       d_n, cap, d_grid_start, d_grid_idx, d_pair_f1, d_pair_f2, min_x, min_y, invW, invH,
                       d_prev_matched, d_matches12, d_nmatches, window_size, nnratio, check_orientation, Ls);
    if (hipGetLastError() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_search_for_initialization launch failed");
    return pg_ctx_scratch_done(c, (hipStream_t)stream);
}

// what the key-frame form (mode 2) takes beyond the common query arrays
struct PgProjKeyFrame { const uint8_t* found; const float* dist3d; const float* minDist; const float* maxDist; float logSf; int orbDist; };

static int pg_search_by_projection_batch(pgorb_ctx* c, int mode, const pgorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int cap,
                                         const int32_t* d_grid_start, const int32_t* d_grid_idx, const int32_t* d_pair_frame, int npairs,
                                         float min_x, float max_x, float min_y, float max_y, const uint8_t* d_kp_has_point, int qcap,
                                         const int32_t* d_nq, const uint8_t* d_valid, const float* d_x, const float* d_y, const int32_t* d_level,
                                         const float* d_aux, const uint8_t* d_qdesc, const uint8_t* d_qobs, float th, float nnratio,
                                         int check_orientation, int32_t* d_assigned, int32_t* d_nmatches, hipStream_t stream,
                                         const PgProjKeyFrame* kf = nullptr)
{
    const bool m2 = mode == 2;
    if (!d_kps || !d_desc || !d_n || cap < 1 || !d_grid_start || !d_grid_idx || npairs < 0 || qcap < 0 ||
        (npairs && (!d_nq || !d_assigned || !d_nmatches)) ||
        (npairs && qcap && (!d_valid || !d_x || !d_y || !d_aux || !d_qdesc || (!m2 && (!d_level || !d_qobs)))) ||
        (m2 && (!kf || (npairs && qcap && (!kf->found || !kf->dist3d || !kf->minDist || !kf->maxDist)) || !(kf->logSf > 0.0f))) ||
        !(max_x > min_x) || !(max_y > min_y))
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_search_by_projection_*");
    if (cap > 16000 || qcap > 16000) return pg_ctx_fail(c, PGORB_E_LIMIT, "more than 16000 keypoints / queries");
    if (!npairs) return 0;
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    PgProjBatch B;
    B.K = d_kps; B.D = d_desc; B.n = d_n; B.cap = cap; B.gstart = d_grid_start; B.gidx = d_grid_idx; B.pairFrame = d_pair_frame;
    B.kpHasPoint = d_kp_has_point; B.qcap = qcap; B.nq = d_nq; B.valid = d_valid; B.x = d_x; B.y = d_y; B.level = d_level; B.aux = d_aux;
    B.desc = d_qdesc; B.hasObs = d_qobs; B.nlevels = pgorb_levels(c); B.th = th;
    B.found = nullptr; B.dist3d = B.minDist = B.maxDist = nullptr; B.logSf = 1.0f; B.orbDist = TH_HIGH; B.maxX = max_x; B.maxY = max_y;
    if (m2) { B.found = kf->found; B.dist3d = kf->dist3d; B.minDist = kf->minDist; B.maxDist = kf->maxDist; B.logSf = kf->logSf; B.orbDist = kf->orbDist; }
    pgorb_scale_tables(c, B.sf, nullptr, nullptr, nullptr);
    const float invW = (float)GRID_COLS / (max_x - min_x), invH = (float)GRID_ROWS / (max_y - min_y);
    // scratch of the two passes: the candidate lists of every query of every pair (PgLists)
    PgLists Ls;
    const size_t szL = pg_lists_layout(nullptr, npairs, std::max(qcap, 1), &Ls);
    void* scratch;
    int rcs = pg_ctx_scratch(c, szL + 256, stream, &scratch);
    if (rcs) return rcs;
    pg_lists_layout(scratch, npairs, std::max(qcap, 1), &Ls);
    const size_t ldsA = (size_t)4 * cap * 2;
    const size_t lds = (size_t)cap * 13 + (size_t)qcap * 10 + 256;
    if (!pg_raise_lds(c, reinterpret_cast<const void*>(k_search_by_projection), 0, lds) ||
        !pg_raise_lds(c, reinterpret_cast<const void*>(k_proj_candidates), 3, ldsA)) return pg_ctx_fail(c, PGORB_E_LIMIT, "SearchByProjection state exceeds the LDS (keypoints * 13 + queries * 10 bytes, 160 KB)");
    if (hipMemsetAsync(Ls.poolTop, 0, (size_t)npairs * 4, stream) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipMemsetAsync failed");
    if (qcap) hipLaunchKernelGGL(k_proj_candidates, dim3((qcap + 3) / 4, npairs), dim3(256), ldsA, stream, B, min_x, min_y, invW, invH, mode, Ls);
    hipLaunchKernelGGL(k_search_by_projection, dim3(npairs), dim3(RR_T), lds, stream, B, min_x, min_y, invW, invH, mode, nnratio,
                       check_orientation, Ls, d_assigned, d_nmatches);
    if (hipGetLastError() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_search_by_projection launch failed");
    return pg_ctx_scratch_done(c, stream);
}

int pgorb_search_by_projection_points_batch_device(pgorb_ctx* c, const pgorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
        int cap_per_frame, const int32_t* d_grid_start, const int32_t* d_grid_idx, const int32_t* d_pair_frame, int npairs,
        float min_x, float max_x, float min_y, float max_y, const uint8_t* d_kp_has_point, int qcap, const int32_t* d_nq,
        const uint8_t* d_valid, const float* d_proj_x, const float* d_proj_y, const int32_t* d_level, const float* d_view_cos,
        const uint8_t* d_point_desc, const uint8_t* d_point_has_obs, float th, float nnratio, int32_t* d_assigned, int32_t* d_nmatches,
        void* stream)
{
    if (!c) return PGORB_E_ARG;
    return pg_search_by_projection_batch(c, 0, d_kps, d_desc, d_n, cap_per_frame, d_grid_start, d_grid_idx, d_pair_frame, npairs, min_x, max_x,
                                         min_y, max_y, d_kp_has_point, qcap, d_nq, d_valid, d_proj_x, d_proj_y, d_level, d_view_cos, d_point_desc,
                                         d_point_has_obs, th, nnratio, 0, d_assigned, d_nmatches, (hipStream_t)stream);
}

int pgorb_search_by_projection_frame_batch_device(pgorb_ctx* c, const pgorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
        int cap_per_frame, const int32_t* d_grid_start, const int32_t* d_grid_idx, const int32_t* d_pair_frame, int npairs,
        float min_x, float max_x, float min_y, float max_y, const uint8_t* d_kp_has_point, int qcap, const int32_t* d_nq,
        const uint8_t* d_valid, const float* d_u, const float* d_v, const int32_t* d_last_octave, const float* d_last_angle,
        const uint8_t* d_point_desc, const uint8_t* d_point_has_obs, float th, int check_orientation, int32_t* d_assigned,
        int32_t* d_nmatches, void* stream)
{
    if (!c) return PGORB_E_ARG;
    return pg_search_by_projection_batch(c, 1, d_kps, d_desc, d_n, cap_per_frame, d_grid_start, d_grid_idx, d_pair_frame, npairs, min_x, max_x,
                                         min_y, max_y, d_kp_has_point, qcap, d_nq, d_valid, d_u, d_v, d_last_octave, d_last_angle, d_point_desc,
                                         d_point_has_obs, th, 0.f, check_orientation, d_assigned, d_nmatches, (hipStream_t)stream);
}

int pgorb_search_by_projection_keyframe_batch_device(pgorb_ctx* c, const pgorb_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
        int cap_per_frame, const int32_t* d_grid_start, const int32_t* d_grid_idx, const int32_t* d_pair_frame, int npairs,
        float min_x, float max_x, float min_y, float max_y, const uint8_t* d_kp_has_point, int qcap, const int32_t* d_nq,
        const uint8_t* d_valid, const uint8_t* d_already_found, const float* d_u, const float* d_v, const float* d_dist3d,
        const float* d_min_distance, const float* d_max_distance, const float* d_kf_angle, const uint8_t* d_point_desc,
        float log_scale_factor, float th, int orb_dist, int check_orientation, int32_t* d_assigned, int32_t* d_nmatches, void* stream)
{
    if (!c) return PGORB_E_ARG;
    const PgProjKeyFrame kf = {d_already_found, d_dist3d, d_min_distance, d_max_distance, log_scale_factor, orb_dist};
    return pg_search_by_projection_batch(c, 2, d_kps, d_desc, d_n, cap_per_frame, d_grid_start, d_grid_idx, d_pair_frame, npairs, min_x, max_x,
                                         min_y, max_y, d_kp_has_point, qcap, d_nq, d_valid, d_u, d_v, nullptr, d_kf_angle, d_point_desc,
                                         nullptr, th, 0.f, check_orientation, d_assigned, d_nmatches, (hipStream_t)stream, &kf);
}

// the contract's logarithm and the Frame's mfLogScaleFactor under it (Frame.cc:188), MapPoint::PredictScale (MapPoint.cc:516-531)
float pgorb_log_f(float x) { return pg_log_f(x); }
float pgorb_log_scale_factor(const pgorb_ctx* c)
{
    if (!c) return 0.0f;
    float sf[PG_MAXL + 1];
    pgorb_scale_tables(c, sf, nullptr, nullptr, nullptr);
    return pg_log_f(sf[1]);                              // mvScaleFactor[1] = (float)(1.0f * (double)scaleFactor) = mfScaleFactor
}
int pgorb_predict_scale(const pgorb_ctx* c, float max_distance, float current_dist)
{
    if (!c) return PGORB_E_ARG;
    return pg_predict_scale(max_distance, current_dist, pgorb_log_scale_factor(c), pgorb_levels(c));
}

// single frame through host buffers: a one-pair batch.  Round 4: the eleven input arrays are packed into the context's
// page-locked bounce buffer and travel as ONE asynchronous upload, the results (assignment array + count) as one
// download -- eleven synchronous pageable hipMemcpy calls were 0.3 ms of a 1.6-ms call.
struct PgProjHostKF { const uint8_t* found; const float* dist3d; const float* minDist; const float* maxDist; float logSf; int orbDist; };
static int pg_search_by_projection_host(pgorb_ctx* c, int mode, const pgorb_keypoint* kps, const uint8_t* desc, int n,
                                        float min_x, float max_x, float min_y, float max_y, const uint8_t* kp_has_point,
                                        int nq, const uint8_t* valid, const float* qx, const float* qy, const int32_t* level,
                                        const float* aux, const uint8_t* qdesc, const uint8_t* qobs, float th, float nnratio,
                                        int check_orientation, int32_t* assigned, const PgProjHostKF* kf = nullptr)
{
    const bool m2 = mode == 2;
    if (n < 0 || nq < 0 || (n && (!kps || !desc || !assigned)) ||
        (nq && (!valid || !qx || !qy || !aux || !qdesc || (!m2 && (!level || !qobs)))) ||
        (m2 && (!kf || (nq && (!kf->found || !kf->dist3d || !kf->minDist || !kf->maxDist)))) ||
        !(max_x > min_x) || !(max_y > min_y))
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_search_by_projection_*");
    for (int i = 0; i < n; i++) assigned[i] = -1;
    if (!n || !nq) return 0;
    if (n > 16000 || nq > 16000) return pg_ctx_fail(c, PGORB_E_LIMIT, "more than 16000 keypoints / queries");
    auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
    // uploaded part first (one copy), device-only scratch and the results behind it
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t oN = place(64), oK = place((size_t)n * sizeof(pgorb_keypoint)), oD = place((size_t)n * 32), oH = place(n), oV = place(nq),
                 oX = place((size_t)nq * 4), oY = place((size_t)nq * 4), oL = place((size_t)nq * 4), oA = place((size_t)nq * 4),
                 oQD = place((size_t)nq * 32), oO = place(nq), oF = place(m2 ? nq : 0), oD3 = place(m2 ? (size_t)nq * 4 : 0),
                 oDmin = place(m2 ? (size_t)nq * 4 : 0), oDmax = place(m2 ? (size_t)nq * 4 : 0);
    const size_t upBytes = off;
    const size_t oGS = place((size_t)(GRID_CELLS + 1) * 4), oGI = place((size_t)n * 4);
    const size_t oAs = place((size_t)n * 4), oR = place(64);
    const size_t total = off, downBytes = total - oAs;
    void *dv, *hv;
    int rc = pg_ctx_stage(c, 0, total, &dv);
    if (rc) return rc;
    if ((rc = pg_ctx_pinned(c, std::max(upBytes, downBytes), &hv))) return rc;
    uint8_t* d = (uint8_t*)dv; uint8_t* h = (uint8_t*)hv;
    const int32_t cnt[2] = {n, nq};
    memcpy(h + oN, cnt, 8);
    memcpy(h + oK, kps, (size_t)n * sizeof(pgorb_keypoint)); memcpy(h + oD, desc, (size_t)n * 32);
    if (kp_has_point) memcpy(h + oH, kp_has_point, n);
    memcpy(h + oV, valid, nq); memcpy(h + oX, qx, (size_t)nq * 4); memcpy(h + oY, qy, (size_t)nq * 4);
    if (level) memcpy(h + oL, level, (size_t)nq * 4);
    memcpy(h + oA, aux, (size_t)nq * 4); memcpy(h + oQD, qdesc, (size_t)nq * 32);
    if (qobs) memcpy(h + oO, qobs, nq);
    if (m2) { memcpy(h + oF, kf->found, nq); memcpy(h + oD3, kf->dist3d, (size_t)nq * 4); memcpy(h + oDmin, kf->minDist, (size_t)nq * 4); memcpy(h + oDmax, kf->maxDist, (size_t)nq * 4); }
    if (hipMemcpyAsync(d, h, upBytes, hipMemcpyHostToDevice, 0) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy H2D failed");
    if ((rc = pgorb_frame_grid_batch_device(c, (pgorb_keypoint*)(d + oK), (int32_t*)(d + oN), 1, n, min_x, max_x, min_y,
                                            max_y, (int32_t*)(d + oGS), (int32_t*)(d + oGI), 0))) return rc;
    const PgProjKeyFrame dkf = {d + oF, (float*)(d + oD3), (float*)(d + oDmin), (float*)(d + oDmax), m2 ? kf->logSf : 1.0f, m2 ? kf->orbDist : 0};
    rc = pg_search_by_projection_batch(c, mode, (pgorb_keypoint*)(d + oK), d + oD, (int32_t*)(d + oN), n, (int32_t*)(d + oGS), (int32_t*)(d + oGI),
                                       nullptr, 1, min_x, max_x, min_y, max_y, kp_has_point ? d + oH : nullptr, nq, (int32_t*)(d + oN) + 1, d + oV,
                                       (float*)(d + oX), (float*)(d + oY), (int32_t*)(d + oL), (float*)(d + oA), d + oQD, d + oO, th, nnratio,
                                       check_orientation, (int32_t*)(d + oAs), (int32_t*)(d + oR), 0, m2 ? &dkf : nullptr);
    if (rc) return rc;
    if (hipMemcpyAsync(h, d + oAs, downBytes, hipMemcpyDeviceToHost, 0) != hipSuccess || hipStreamSynchronize(0) != hipSuccess)
        return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy D2H failed");
    memcpy(assigned, h, (size_t)n * 4);
    int32_t nm;
    memcpy(&nm, h + (oR - oAs), 4);
    return nm;
}

int pgorb_search_by_projection_keyframe(pgorb_ctx* c, const pgorb_keypoint* kps, const uint8_t* desc, int n, float min_x,
                                        float max_x, float min_y, float max_y, const uint8_t* kp_has_point, int npoints,
                                        const uint8_t* valid, const uint8_t* already_found, const float* u, const float* v,
                                        const float* dist3d, const float* min_distance, const float* max_distance,
                                        const float* kf_angle, const uint8_t* point_desc, float log_scale_factor, float th,
                                        int orb_dist, int check_orientation, int32_t* assigned)
{
    if (!c) return PGORB_E_ARG;
    if (!(log_scale_factor > 0.0f)) return pg_ctx_fail(c, PGORB_E_ARG, "pgorb_search_by_projection_keyframe: log_scale_factor must be positive");
    const PgProjHostKF kf = {already_found, dist3d, min_distance, max_distance, log_scale_factor, orb_dist};
    return pg_search_by_projection_host(c, 2, kps, desc, n, min_x, max_x, min_y, max_y, kp_has_point, npoints, valid, u, v, nullptr,
                                        kf_angle, point_desc, nullptr, th, 0.f, check_orientation, assigned, &kf);
}

int pgorb_search_by_projection_points(pgorb_ctx* c, const pgorb_keypoint* kps, const uint8_t* desc, int n, float min_x,
                                      float max_x, float min_y, float max_y, const uint8_t* kp_has_point, int npoints,
                                      const uint8_t* valid, const float* proj_x, const float* proj_y, const int32_t* level,
                                      const float* view_cos, const uint8_t* point_desc, const uint8_t* point_has_obs,
                                      float th, float nnratio, int32_t* assigned)
{
    if (!c) return PGORB_E_ARG;
    return pg_search_by_projection_host(c, 0, kps, desc, n, min_x, max_x, min_y, max_y, kp_has_point, npoints, valid, proj_x, proj_y,
                                        level, view_cos, point_desc, point_has_obs, th, nnratio, 0, assigned);
}

int pgorb_search_by_projection_frame(pgorb_ctx* c, const pgorb_keypoint* kps, const uint8_t* desc, int n, float min_x,
                                     float max_x, float min_y, float max_y, const uint8_t* kp_has_point, int nlast,
                                     const uint8_t* valid, const float* u, const float* v, const int32_t* last_octave,
                                     const float* last_angle, const uint8_t* point_desc, const uint8_t* point_has_obs,
                                     float th, int check_orientation, int32_t* assigned)
{
    if (!c) return PGORB_E_ARG;
    return pg_search_by_projection_host(c, 1, kps, desc, n, min_x, max_x, min_y, max_y, kp_has_point, nlast, valid, u, v,
                                        last_octave, last_angle, point_desc, point_has_obs, th, 0.f, check_orientation, assigned);
}

int pgorb_frame_grid(pgorb_ctx* c, const pgorb_keypoint* kps, int n, float min_x, float max_x, float min_y,
                     float max_y, int32_t* grid_start, int32_t* grid_idx)
{
    if (!c) return PGORB_E_ARG;
    if (n < 0 || (n && (!kps || !grid_idx)) || !grid_start) return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_frame_grid");
    const int cap = n > 0 ? n : 1;
    // [n | keypoints] go up in one copy, [grid_start | grid_idx] come back in one
    const size_t oK = 64, oS = oK + (((size_t)cap * sizeof(pgorb_keypoint) + 63) & ~(size_t)63);
    const size_t oI = oS + (size_t)(GRID_CELLS + 1) * 4, total = oI + (size_t)cap * 4;
    void *dv, *hv;
    int rc = pg_ctx_stage(c, 0, total, &dv);
    if (rc) return rc;
    if ((rc = pg_ctx_pinned(c, std::max(oS, total - oS), &hv))) return rc;
    uint8_t* d = (uint8_t*)dv; uint8_t* h = (uint8_t*)hv;
    memcpy(h, &n, 4);
    if (n) memcpy(h + oK, kps, (size_t)n * sizeof(pgorb_keypoint));
    if (hipMemcpyAsync(d, h, oS, hipMemcpyHostToDevice, 0) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy failed");
    if ((rc = pgorb_frame_grid_batch_device(c, (pgorb_keypoint*)(d + oK), (int32_t*)d, 1, cap, min_x, max_x, min_y, max_y,
                                            (int32_t*)(d + oS), (int32_t*)(d + oI), 0))) return rc;
    if (hipMemcpyAsync(h, d + oS, total - oS, hipMemcpyDeviceToHost, 0) != hipSuccess || hipStreamSynchronize(0) != hipSuccess)
        return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy failed");
    memcpy(grid_start, h, (size_t)(GRID_CELLS + 1) * 4);
    if (n) memcpy(grid_idx, h + (oI - oS), (size_t)n * 4);
    return 0;
}

int pgorb_search_for_initialization(pgorb_ctx* c, const pgorb_keypoint* kps1, const uint8_t* desc1, int n1,
                                    const pgorb_keypoint* kps2, const uint8_t* desc2, int n2,
                                    float min_x, float max_x, float min_y, float max_y, float* prev_matched,
                                    int32_t* matches12, int window_size, float nnratio, int check_orientation)
{
    if (!c) return PGORB_E_ARG;
    if (n1 < 0 || n2 < 0 || (n1 && (!kps1 || !desc1 || !prev_matched || !matches12)) || (n2 && (!kps2 || !desc2)))
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_search_for_initialization");
    if (n1 == 0) return 0;
    const int cap = (n1 > n2 ? n1 : n2) > 0 ? (n1 > n2 ? n1 : n2) : 1;
    // One slab, laid out so that everything the host supplies is one upload and everything it reads back one
    // download: [misc | kps x2 | desc x2 | prev | matches | nmatches] then the device-only grids.
    auto up64 = [](size_t v) { return (v + 63) & ~(size_t)63; };
    const size_t oMisc = 0;
    const size_t oK = 64;
    const size_t oD = oK + up64((size_t)2 * cap * sizeof(pgorb_keypoint));
    const size_t oP = oD + up64((size_t)2 * cap * 32);
    const size_t oM = oP + up64((size_t)cap * 8);
    const size_t oNm = oM + up64((size_t)cap * 4);
    const size_t oGS = oNm + 64;
    const size_t oGI = oGS + up64((size_t)2 * (GRID_CELLS + 1) * 4);
    const size_t total = oGI + up64((size_t)2 * cap * 4);
    const size_t upBytes = oM, downBytes = oGS - oP;
    void *dv, *hv;
    int rc = pg_ctx_stage(c, 0, total, &dv);
    if (rc) return rc;
    if ((rc = pg_ctx_pinned(c, std::max(upBytes, downBytes), &hv))) return rc;
    uint8_t* d = (uint8_t*)dv; uint8_t* h = (uint8_t*)hv;
    const int32_t misc[5] = {n1, n2, 0, 1, 0};   // n[2], f1, f2
    memcpy(h + oMisc, misc, sizeof(misc));
    memcpy(h + oK, kps1, (size_t)n1 * sizeof(pgorb_keypoint));
    memcpy(h + oD, desc1, (size_t)n1 * 32);
    if (n2) {
        memcpy(h + oK + (size_t)cap * sizeof(pgorb_keypoint), kps2, (size_t)n2 * sizeof(pgorb_keypoint));
        memcpy(h + oD + (size_t)cap * 32, desc2, (size_t)n2 * 32);
    }
    memcpy(h + oP, prev_matched, (size_t)n1 * 8);
    if (hipMemcpyAsync(d, h, upBytes, hipMemcpyHostToDevice, 0) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy H2D failed");
    pgorb_keypoint* dk = (pgorb_keypoint*)(d + oK);
    int32_t* dmisc = (int32_t*)(d + oMisc);
    if ((rc = pgorb_frame_grid_batch_device(c, dk, dmisc, 2, cap, min_x, max_x, min_y, max_y, (int32_t*)(d + oGS), (int32_t*)(d + oGI), 0))) return rc;
    if ((rc = pgorb_search_for_initialization_batch_device(c, dk, d + oD, dmisc, cap, (int32_t*)(d + oGS), (int32_t*)(d + oGI), dmisc + 2,
                                                           dmisc + 3, 1, min_x, max_x, min_y, max_y, (float*)(d + oP), (int32_t*)(d + oM),
                                                           (int32_t*)(d + oNm), window_size, nnratio, check_orientation, 0))) return rc;
    if (hipMemcpyAsync(h, d + oP, downBytes, hipMemcpyDeviceToHost, 0) != hipSuccess || hipStreamSynchronize(0) != hipSuccess)
        return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy D2H failed");
    memcpy(prev_matched, h, (size_t)n1 * 8);
    memcpy(matches12, h + (oM - oP), (size_t)n1 * 4);
    int32_t nm;
    memcpy(&nm, h + (oNm - oP), 4);
    return nm;
}

}  // extern "C"

// api.hip -- host side of libpgorb.so: context, per-frame-size plan, launches, C ABI.
//
// Mirrors the host logic of ORB_SLAM2::ORBextractor that is not per-pixel work:
//   constructor tables      thirdparty/orb-slam2/src/ORBextractor.cc:410-470
//   level sizes             ORBextractor.cc:1110-1111
//   cell grid               ORBextractor.cc:773-787
//   quadtree roots          ORBextractor.cc:543-545
//   operator() sequencing   ORBextractor.cc:1042-1104
// Everything per-pixel / per-keypoint runs in the HIP kernels; there is no CPU fallback.
#include "pgorb_internal.h"

#include <chrono>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct Arena {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct pgorb_ctx {
    pgorb_params prm;
    double scaleFactor;                       // the reference keeps a double member
    float mvScaleFactor[PG_MAXL + 1], mvInvScaleFactor[PG_MAXL + 1];
    float mvLevelSigma2[PG_MAXL + 1], mvInvLevelSigma2[PG_MAXL + 1];
    int mnFeaturesPerLevel[PG_MAXL + 1];
    std::string err;
    // plan for the current frame size
    PgPlan plan;
    int planW = 0, planH = 0, planBatch = 0;
    bool planValid = false;
    // device memory
    Arena pyr, cand, sel, nodes, counters, tables, cellCand, cellCount, cellTab, cellTabBal, qtTab, qtLeaf;
    int qtThreads = 0;                        // K3 threads per workgroup: 0 = per launch (pgorb_set_option "quadtree_threads")
    int qtSplit = 2;                          // K3's candidate pass as its own launch: 0 no, 1 yes, 2 by frame size and batch (pgorb_set_option "quadtree_split")
    PgFusePlan fuse;                          // tables of the fused launches (make_plan)
    int fused = 0;                            // pgorb_set_option "fused_levels": 1 = resize + detect in one launch per level (fused.hip; measured slower, off by default)
    int fastTilePitch = 0, fastWpb = 1, fastCpw = PG_FAST_CPW_DEFAULT;       // K2 tile-shape sweep (pgorb_set_option "fast_tile_pitch" / "fast_waves_per_block")
    // K1 beside K2 (pgorb_set_option "pipeline_pyramid"): the pyramid chain on a high-priority side stream, K2 level by
    // level on a second one as the levels appear
    int pipePyr = 0;
    hipStream_t sPyr = nullptr, sFast = nullptr;
    hipEvent_t evFork = nullptr, evLevel[PG_MAXL] = {}, evPyrDone = nullptr, evFastDone = nullptr;
    // K3 / K4-6 of a group of levels beside K2 of the next group (pgorb_set_option "pipeline_levels", bit l = a group
    // starts at level l): K3 is one workgroup's critical path per (frame, level) and leaves the chip mostly idle
    int pipeLev = 0, pipeLevPrio = 0;
    // a device-resident stream's lanes (pgorb_stream_create_device): recorded behind the batch's last pyramid launch / behind K2,
    // so that the NEXT batch (on another lane) can be held back until this one has reached that point (stagger)
    hipEvent_t evPyrEnd = nullptr, evFastEnd = nullptr;
    hipStream_t sQt = nullptr, sDesc = nullptr;
    hipEvent_t evGrpFast[PG_MAXL] = {}, evGrpQt[PG_MAXL] = {}, evDescDone = nullptr;
    Arena stageA, stageB, stageOut, stageSfi, vocab;
    Arena xdesc;                              // matcher scratch: train descriptors as +-1 bytes (match.hip, match_mode 0 only)
    PgMatchOpts mx;                           // this context's matcher settings (pgorb_set_option "matcher" / "match_mode")
    void* pinned = nullptr;                   // page-locked bounce buffer for bulk result download
    size_t pinnedBytes = 0;
    uint8_t* pinIn = nullptr;                 // page-locked input staging of the host-frame calls (pgorb_extract*)
    size_t pinInBytes = 0;
    int chunkBytes = 512 << 10, noStage = 1;  // PGORB_EXTRACT_STAGE=1: frames through the context's page-locked staging buffer in chunks of this size (measured SLOWER
                                              // than the runtime's own pageable-copy path, DESIGN.md section 6 round 4: kept as the A/B switch)
    Arena outBlk;                             // status word | counts | keypoints | descriptors of a host-frame call: one download
    double hostUs[4] = {0, 0, 0, 0}; int hostCalls = 0;
    // the host-frame calls run on a stream of the context's own, and replay their kernel chain (K1..K6 + the result download)
    // as a HIP graph from the second call with the same plan / batch size on (PGORB_EXTRACT_NO_GRAPH=1: direct launches)
    hipStream_t sHost = nullptr;
    int useGraph = 1, planEpoch = 0;
    struct HostGraph { hipGraph_t g = nullptr; hipGraphExec_t exec = nullptr; int nframes = 0, epoch = -1, seenFrames = 0, seenEpoch = -1; void* pinned = nullptr; size_t outBytes = 0; } hg;
    int vocabK = 0, vocabL = 0, vocabNodes = 0;
    int lastFrames = 0;
    bool lastAliased = false;
    // stage profiling (HIP events on the launch stream)
    std::vector<hipEvent_t> evExtract;        // 5 per armed extract call
    std::vector<hipEvent_t> evMatch;          // 2 per armed match call
    int profMax = 0, profExtract = 0, profMatch = 0;
    std::vector<pgorb_stream*> streams;       // live pgorb_stream_* objects of this context (pgorb_destroy takes them along)
    // the matchers' shared scratch arena (stageSfi) may be used from different caller streams: the last use is an event
    hipEvent_t evSfi = nullptr; hipStream_t sfiStream = nullptr; bool sfiUsed = false;
};

namespace {

int fail(pgorb_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define PG_HIP(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((c), PGORB_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));   \
    } while (0)

int ensure(pgorb_ctx* c, Arena& a, size_t bytes)
{
    if (a.bytes >= bytes && a.p) return 0;
    if (a.p) { (void)hipFree(a.p); a.p = nullptr; a.bytes = 0; }
    bytes = (bytes + 4095) & ~(size_t)4095;
    PG_HIP(c, hipMalloc(&a.p, bytes));
    a.bytes = bytes;
    return 0;
}

inline int cvRound(double v) { return (int)lrint(v); }
inline int cvFloor(double v) { int i = (int)v; return i - (v < i); }

// level sizes, ORBextractor.cc:1110-1111
void level_size(const pgorb_ctx* c, int level, int w, int h, int* lw, int* lh)
{
    const float scale = c->mvInvScaleFactor[level];
    *lw = cvRound((double)((float)w * scale));
    *lh = cvRound((double)((float)h * scale));
}

struct LevelGeom { int w, h, nCols, nRows, wCell, hCell, nIni; float hX; };

int level_geometry(const pgorb_ctx* c, int w, int h, LevelGeom* g)
{
    for (int l = 0; l < c->prm.nlevels; l++) {
        level_size(c, l, w, h, &g[l].w, &g[l].h);
        const float width = (float)(g[l].w - 2 * PG_EDGE), height = (float)(g[l].h - 2 * PG_EDGE);
        const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);     // :784-785
        if (nCols < 1 || nRows < 1) return PGORB_E_TOOSMALL;
        g[l].nCols = nCols; g[l].nRows = nRows;
        g[l].wCell = (int)ceilf(width / nCols);                                   // :786-787
        g[l].hCell = (int)ceilf(height / nRows);
        const int rw = g[l].w - 2 * PG_EDGE, rh = g[l].h - 2 * PG_EDGE;
        g[l].nIni = (int)roundf((float)rw / rh);                                  // :543
        // nIni == 0 (region more than twice as tall as wide): the reference divides by it and then
        // indexes an empty vector -- but only when the level has candidates (:545-570); with none it
        // returns an empty level.  Same here: K3 raises PGORB_E_TOOSMALL iff such a level has candidates.
        if (g[l].nIni < 1) g[l].nIni = 0;
        g[l].hX = g[l].nIni ? (float)rw / g[l].nIni : 0.f;                        // :545
    }
    return 0;
}

// cv::resize INTER_LINEAR coefficient set-up for 8U (OpenCV 2.4 imgwarp.cpp; Appendix A1)
void build_resize_tables(int sw, int sh, int dw, int dh, std::vector<int32_t>& xofs,
                         std::vector<int32_t>& xofs1, std::vector<int16_t>& xalpha,
                         std::vector<int32_t>& yofs, std::vector<int16_t>& ybeta)
{
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    xofs.resize(dw); xofs1.resize(dw); xalpha.resize(2 * dw); yofs.resize(2 * dh); ybeta.resize(2 * dh);
    auto sat16 = [](int v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); };
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }          // single-tap columns (dx >= xmax)
        xofs[dx] = sx;
        xofs1[dx] = sx + 1 < sw ? sx + 1 : sw - 1;
        xalpha[2 * dx] = sat16(cvRound((double)((1.f - fx) * 2048)));
        xalpha[2 * dx + 1] = sat16(cvRound((double)(fx * 2048)));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        int r0 = sy, r1 = sy + 1;
        r0 = r0 < 0 ? 0 : (r0 >= sh ? sh - 1 : r0);
        r1 = r1 < 0 ? 0 : (r1 >= sh ? sh - 1 : r1);
        yofs[2 * dy] = r0; yofs[2 * dy + 1] = r1;
        ybeta[2 * dy] = sat16(cvRound((double)((1.f - fy) * 2048)));
        ybeta[2 * dy + 1] = sat16(cvRound((double)(fy * 2048)));
    }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Build (or reuse) the plan for w x h frames and a batch of `nframes`.
int make_plan(pgorb_ctx* c, int w, int h, int nframes)
{
    if (w > c->prm.max_width || h > c->prm.max_height)
        return fail(c, PGORB_E_LIMIT, "frame %dx%d exceeds context maximum %dx%d", w, h,
                    c->prm.max_width, c->prm.max_height);
    if (nframes > c->prm.max_batch)
        return fail(c, PGORB_E_LIMIT, "batch %d exceeds max_batch %d", nframes, c->prm.max_batch);
    if (c->planValid && c->planW == w && c->planH == h) return 0;

    const int L = c->prm.nlevels, B = c->prm.max_batch;
    LevelGeom g[PG_MAXL];
    int rc = level_geometry(c, w, h, g);
    if (rc) return fail(c, rc, "frame %dx%d too small: every pyramid level needs >= 62 px per side "
                               "and aspect >= 0.5", w, h);
    // the old plan dies here: a failure below (allocation, copy, limit) must not leave planValid set for
    // arenas that were already resized or a plan that is half written.  Work still in flight on any stream
    // (pgorb_stream_* batches, *_device calls on the caller's stream) reads the old tables and arenas: drain it
    // before they are overwritten.
    c->planValid = false;
    PG_HIP(c, hipDeviceSynchronize());
    PgPlan& P = c->plan;
    memset(&P, 0, sizeof(P));
    memset(&c->fuse, 0, sizeof(c->fuse));
    P.nlevels = L; P.iniTh = c->prm.ini_th_fast; P.minTh = c->prm.min_th_fast;
    P.tieMode = c->prm.blur_tie_mode;

    // --- resize tables ---
    std::vector<uint8_t> tab;
    struct TabOff { size_t xofs, xofs1, xalpha, yofs, ybeta, qtab, yrel, qtab2, rowgrp, tilex, fbands, fcols; int cpr, prows, fnb, fspb, frows; bool hasQ, hasY, hasQ2; } toff[PG_MAXL];
    for (int l = 0; l < PG_MAXL; l++) toff[l].fnb = 0;
    int pyrGpw = 2;                                                   // 4-row groups per wave of the LDS-staged resize
    if (const char* e = getenv("PGORB_PYR_TILE_ROWS")) { const int r = atoi(e); if (r == 16 || r == 32 || r == 64) pyrGpw = r / 16; }
    for (int l = 1; l < L; l++) {
        std::vector<int32_t> xo, xo1, yo; std::vector<int16_t> xa, yb;
        build_resize_tables(g[l - 1].w, g[l - 1].h, g[l].w, g[l].h, xo, xo1, xa, yo, yb);
        auto put = [&](const void* p, size_t n) {
            size_t off = align_up(tab.size(), 16);
            tab.resize(off + n);
            memcpy(tab.data() + off, p, n);
            return off;
        };
        toff[l].xofs = put(xo.data(), xo.size() * 4);
        toff[l].xofs1 = put(xo1.data(), xo1.size() * 4);
        toff[l].xalpha = put(xa.data(), xa.size() * 2);
        toff[l].yofs = put(yo.data(), yo.size() * 4);
        toff[l].ybeta = put(yb.data(), yb.size() * 2);
        // quad table for the fast path
        const int nq = (g[l].w + 3) / 4;
        std::vector<PgQuadTab> qt(nq);
        bool ok = true;
        for (int q = 0; q < nq; q++) {
            PgQuadTab& T = qt[q];
            memset(&T, 0, sizeof(T));
            T.base_dw = xo[4 * q] >> 2;
            for (int k = 0; k < 4; k++) {
                const int dx = 4 * q + k;
                if (dx >= g[l].w) break;
                const int o = xo[dx] - 4 * T.base_dw;
                if (o < 0 || o > 8) ok = false;
                // tap 1 is read at o+1; where cv::resize clamps it (xofs1 == xofs) its weight is 0
                if (xo1[dx] != xo[dx] + 1 && xa[2 * dx + 1] != 0) ok = false;
                T.offs |= (uint32_t)(o & 15) << (4 * k);
                T.a0[k] = xa[2 * dx]; T.a1[k] = xa[2 * dx + 1];
            }
        }
        toff[l].hasQ = ok;
        toff[l].qtab = put(qt.data(), qt.size() * sizeof(PgQuadTab));
        // 8-byte-window table: xb = first tap of the quad (clamped so the window stays inside the row)
        std::vector<PgQuadTab2> q2(nq);
        bool ok2 = g[l - 1].w >= 8;
        for (int q = 0; q < nq && ok2; q++) {
            PgQuadTab2& T = q2[q];
            memset(&T, 0, sizeof(T));
            T.xb = std::min(xo[4 * q], g[l - 1].w - 8);
            for (int k = 0; k < 4; k++) {
                const int dx = 4 * q + k;
                if (dx >= g[l].w) { T.sel[k] = 0x0c0c0c0cu; continue; }
                const int o0 = xo[dx] - T.xb, o1 = xo1[dx] - T.xb;
                if (o0 < 0 || o0 > 7 || o1 < 0 || o1 > 7) { ok2 = false; break; }
                T.sel[k] = (uint32_t)o0 | (0x0cu << 8) | ((uint32_t)o1 << 16) | (0x0cu << 24);
                T.coef[k] = (uint32_t)(uint16_t)xa[2 * dx] | ((uint32_t)(uint16_t)xa[2 * dx + 1] << 16);
                if (xa[2 * dx] < 0 || xa[2 * dx + 1] < 0) ok2 = false;
            }
        }
        toff[l].hasQ2 = ok2;
        toff[l].qtab2 = put(q2.data(), q2.size() * sizeof(PgQuadTab2));
        // row pattern for the 4-rows-per-lane kernel: r0(dy) - r0(group base) - d in {0,1}, r1 - r0 in {0,1}
        std::vector<uint8_t> yr(g[l].h, 0);
        bool oky = true;
        for (int dy = 0; dy < g[l].h; dy++) {
            const int base = yo[2 * (dy & ~3)], d = dy & 3;
            const int e0 = yo[2 * dy] - base - d, e1 = yo[2 * dy + 1] - yo[2 * dy];
            if (e0 < 0 || e0 > 1 || e1 < 0 || e1 > 1) oky = false;
            yr[dy] = (uint8_t)((e0 & 1) | ((e1 & 1) << 1));
        }
        toff[l].hasY = oky;
        toff[l].yrel = put(yr.data(), yr.size());
        std::vector<PgRowGrp> rg((g[l].h + 3) / 4 + 4);               // + records of slack (a wave loads its 1, 2 or 4 records at once)
        for (size_t gi = 0; gi < rg.size(); gi++) {
            PgRowGrp& R = rg[gi];
            memset(&R, 0, sizeof(R));
            R.sFirst = yo[2 * std::min((int)(4 * gi), g[l].h - 1)];
            for (int d = 0; d < 4; d++) {
                const int dy = std::min((int)(4 * gi) + d, g[l].h - 1);
                R.yrel4 |= (uint32_t)yr[dy] << (8 * d);
                R.ybeta[2 * d] = yb[2 * dy]; R.ybeta[2 * d + 1] = yb[2 * dy + 1];
            }
        }
        toff[l].rowgrp = put(rg.data(), rg.size() * sizeof(PgRowGrp));
        // LDS-staged variant of the 4x4 kernel: a 256 x 32 destination tile stages its source
        // rectangle (16-byte chunks from a 16-aligned first column) through LDS
        toff[l].cpr = 0; toff[l].prows = 0;
        {
            const int tileRows = 16 * pyrGpw, gpt = 4 * pyrGpw;         // rows / 4-row groups per tile
            const int ntx = (g[l].w + 255) / 256, nty = (g[l].h + tileRows - 1) / tileRows, ngrp = (g[l].h + 3) / 4;
            std::vector<int32_t> tx0(ntx, 0);
            int span = 0, rows = 0;
            if (ok2 && oky) {
                for (int tx = 0; tx < ntx; tx++) {
                    const int qf = 64 * tx, ql = std::min(64 * tx + 63, nq - 1);
                    tx0[tx] = q2[qf].xb & ~15;
                    span = std::max(span, q2[ql].xb + 8 - tx0[tx]);
                }
                for (int ty = 0; ty < nty; ty++) {
                    const int gf = gpt * ty, gl = std::min(gpt * ty + gpt - 1, ngrp - 1);
                    rows = std::max(rows, rg[gl].sFirst + 6 - rg[gf].sFirst);
                }
                const int cpr = (span + 4 + 15) / 16;
                if (cpr <= 32 && (size_t)rows * cpr * 16 <= (size_t)10 * 1024 * pyrGpw) { toff[l].cpr = cpr; toff[l].prows = rows; }
            }
            toff[l].tilex = put(tx0.data(), tx0.size() * 4);
        }
        // fused resize + detect (fused.hip) with level l - 1 as the source: bands = the source level's cell rows (hCell + 6 staged
        // rows from row 16 + i hCell) between an edge band above (rows 0 ...) and below; a band owns the 4-row groups of level l whose
        // first source row lies in [its first row, the next band's first row), so that the six source rows of every group are staged;
        // tile columns = NS cells, a column owns the quads whose window starts in [15 + first cell * wCell, ... of the next column).
        toff[l].fnb = 0;
        {
            const LevelGeom& S = g[l - 1];
            const int ngrp = (g[l].h + 3) / 4, rowsB = S.hCell + 6;
            bool fok = ok2 && oky && S.wCell <= 32 && S.hCell <= 40 && S.hCell >= 16 && S.wCell >= 16;       // (the detector's narrow form; 3 x 21 staged rows)
            std::vector<int32_t> bands, cols;
            int nb = 0;
            if (fok) {
                const int maxS = rg[ngrp - 1].sFirst;
                std::vector<int> Y{0};
                for (int i = 0; i < S.nRows || PG_EDGE + i * S.hCell <= maxS; i++) Y.push_back(PG_EDGE + i * S.hCell);
                int gi = 0;
                for (size_t b = 0; b < Y.size(); b++) {
                    const int yNext = b + 1 < Y.size() ? Y[b + 1] : INT32_MAX;
                    const int g0 = gi;
                    while (gi < ngrp && rg[gi].sFirst < yNext) {
                        if (rg[gi].sFirst < Y[b] || rg[gi].sFirst + 5 - Y[b] > rowsB - 1) fok = false;
                        gi++;
                    }
                    bands.insert(bands.end(), {g0, gi - g0});          // (the kernel derives the band's first row and cell row from b)
                    nb++;
                }
                if (gi != ngrp) fok = false;
            }
            int spbF = 0;
            if (fok) {
                // slot columns: s = -1 the left edge's pseudo-cell (source columns from 0), s >= 0 from column 15 + s wCell (the cell's
                // iniX - 1; past the last cell column: pseudo-cells that only serve the resize); a quad belongs to the column its
                // 8-byte window starts in -- and then lies inside that column's 48 bytes
                auto slotOf = [&](int xb) { return xb < PG_EDGE - 1 ? -1 : (xb - (PG_EDGE - 1)) / S.wCell; };
                auto slotX = [&](int sl) { return sl < 0 ? 0 : PG_EDGE - 1 + sl * S.wCell; };
                int nSlotsTotal = S.nCols;
                for (int q = 0; q < nq; q++) {
                    const int sl = slotOf(q2[q].xb), off = q2[q].xb - slotX(sl);
                    if (q > 0 && q2[q].xb < q2[q - 1].xb) fok = false;
                    if (off < 0 || (off & ~3) + 12 > 48) fok = false;
                    nSlotsTotal = std::max(nSlotsTotal, sl + 1);
                }
                spbF = nSlotsTotal + 1;
                int qi = 0, maxNq = 0, maxNg = 0;
                for (int sl = -1; sl < nSlotsTotal; sl++) {
                    const int q0 = qi;
                    while (qi < nq && slotOf(q2[qi].xb) <= sl) qi++;
                    const int n = qi - q0, magic = n > 0 ? 65536 / n + 1 : 65536;
                    for (int lane = 0; lane < 64 && n > 0; lane++) if (((lane * magic) >> 16) != lane / n) fok = false;
                    maxNq = std::max(maxNq, n);
                    cols.insert(cols.end(), {q0, n, magic, 0});
                }
                if (qi != nq) fok = false;
                for (int b = 0; b < nb; b++) maxNg = std::max(maxNg, bands[2 * b + 1]);
                if (maxNq * maxNg > 64) fok = false;                   // lane = (quad, group)
            }
            if (fok) {
                toff[l].fbands = put(bands.data(), bands.size() * 4);
                toff[l].fcols = put(cols.data(), cols.size() * 4);
                toff[l].fnb = nb; toff[l].fspb = spbF; toff[l].frows = rowsB;
            }
        }
    }
    if ((rc = ensure(c, c->tables, tab.size() + 16))) return rc;
    if (!tab.empty()) PG_HIP(c, hipMemcpy(c->tables.p, tab.data(), tab.size(), hipMemcpyHostToDevice));

    // --- arenas ---
    size_t pyrFrame = 0, candFrame = 0, selFrame = 0, nodeFrame = 0, cellCandFrame = 0;
    int cells = 0, selTotal = 0;
    size_t pyrOff[PG_MAXL];
    for (int l = 0; l < L; l++) {
        PgLevel& V = P.lvl[l];
        V.w = g[l].w; V.h = g[l].h;
        V.pitch = (int)align_up(V.w, 64);
        pyrOff[l] = pyrFrame;
        pyrFrame += align_up((size_t)V.pitch * V.h + 64, 256);
        V.nCols = g[l].nCols; V.nRows = g[l].nRows; V.wCell = g[l].wCell; V.hCell = g[l].hCell;
        V.cellBase = cells; cells += V.nCols * V.nRows;
        V.quota = c->mnFeaturesPerLevel[l];
        V.nIni = g[l].nIni; V.hX = g[l].hX;
        V.selCap = std::max(V.quota + 2, 4 * V.nIni);
        V.nodeCap = (int)align_up(V.selCap + 8, 4);
        // NMS survivors are never 8-adjacent: at most ceil(IW/2)*ceil(IH/2) per cell
        V.candCap = ((V.w - 2 * PG_EDGE) / 2 + V.nCols + 1) * ((V.h - 2 * PG_EDGE) / 2 + V.nRows + 1);
        V.candOff = (int64_t)candFrame; candFrame += align_up(V.candCap, 64);
        V.cellCap = ((V.wCell + 1) / 2) * ((V.hCell + 1) / 2);
        V.cellCandOff = (int64_t)cellCandFrame;
        cellCandFrame += align_up((size_t)V.cellCap * V.nCols * V.nRows, 64);
        V.selOff = (int64_t)selFrame; selFrame += V.selCap;          // slab offset == capacity prefix: K4-6's slot s of a frame IS entry s
        // node arrays of the quadtree: LDS when 30 ints per node fit its 140 KB, else a global slab
        // (quotas above ~1180 keypoints on one level: slower, but no configuration is refused)
        V.nodeOff = -1;
        // K3 keeps a level's node arrays (28 ints per node) in LDS beside its count pyramid (17 KB static) and its aux area
        // (max(pyramid leaves <= 3072, cells of the largest level + 1) ints); what does not fit 160 KB goes to a global slab
        const size_t qtAux = std::max((size_t)3072, (size_t)g[0].nCols * g[0].nRows + 1);
        const size_t qtTab = (size_t)2 * ((g[0].w - 2 * PG_EDGE) + (g[0].h - 2 * PG_EDGE));      // the prologue's coordinate tables share the node area
        if (std::max((size_t)V.nodeCap * 28, qtTab) * sizeof(int) + qtAux * sizeof(int) + 17 * 1024 > 160 * 1024) {
            V.nodeOff = (int64_t)nodeFrame;
            nodeFrame += align_up((size_t)V.nodeCap * 30, 64);
        }
        V.scale = c->mvScaleFactor[l];
        V.patchSize = (float)(int)(31 * c->mvScaleFactor[l]);                     // :836
        selTotal += V.selCap;
        if ((size_t)(V.nCols * V.nRows + 1) * sizeof(int) > 140 * 1024)
            return fail(c, PGORB_E_LIMIT, "too many cells on one level for the quadtree kernel's LDS budget");
        if (V.w > 4095 + 2 * PG_EDGE || V.h > 4095 + 2 * PG_EDGE)
            return fail(c, PGORB_E_LIMIT, "level larger than 4095 px is not supported");
        if (V.selCap > 65535)      // K3's dispatch-order sort packs a keypoint's arrival index into 16 bits
            return fail(c, PGORB_E_LIMIT, "more than 65533 keypoints on one pyramid level are not supported");
    }
    P.totalCells = cells; P.selTotal = selTotal;
    P.cellCandFrame = (int64_t)cellCandFrame;
    P.candFrame = (int64_t)candFrame; P.selFrame = (int64_t)selFrame; P.nodeFrame = (int64_t)nodeFrame;
    if ((rc = ensure(c, c->pyr, pyrFrame * B))) return rc;
    if ((rc = ensure(c, c->cand, candFrame * 8 * B))) return rc;          // uint2 key records
    if ((rc = ensure(c, c->cellCand, cellCandFrame * 4 * B + 64))) return rc;      // (+64: K3's pass reads a cell's slots 16 bytes at a time)
    if ((rc = ensure(c, c->cellCount, (size_t)cells * 4 * B + 64))) return rc;
    selFrame = align_up(selFrame, 16);
    P.selFrame = (int64_t)selFrame;
    if ((rc = ensure(c, c->sel, selFrame * 32 * B))) return rc;           // one 32-byte selection record per keypoint (quadtree.hip, PgSelRec)
    if ((rc = ensure(c, c->nodes, nodeFrame * 4 * B + 64))) return rc;
    if ((rc = ensure(c, c->counters, (size_t)B * PG_MAXL * 4 * 2 + 64))) return rc;
    // per-level counts: K3 writes the entries of the plan's levels in every batch, the others stay zero from here on (K4-6 sums all 16)
    PG_HIP(c, hipMemset(c->counters.p, 0, (size_t)B * PG_MAXL * 4 * 2 + 64));
    for (int l = 0; l < L; l++) {
        PgLevel& V = P.lvl[l];
        V.img = (uint8_t*)c->pyr.p + pyrOff[l] * B;        // level-major: frames of a level adjacent
        V.fstride = (int64_t)align_up((size_t)V.pitch * V.h + 64, 256);
        if (l >= 1) {
            const uint8_t* t = (const uint8_t*)c->tables.p;
            V.xofs = (const int32_t*)(t + toff[l].xofs);
            V.xofs1 = (const int32_t*)(t + toff[l].xofs1);
            V.xalpha = (const int16_t*)(t + toff[l].xalpha);
            V.yofs = (const int32_t*)(t + toff[l].yofs);
            V.ybeta = (const int16_t*)(t + toff[l].ybeta);
            V.qtab = toff[l].hasQ ? (const PgQuadTab*)(t + toff[l].qtab) : nullptr;
            V.yrel = toff[l].hasY ? (t + toff[l].yrel) : nullptr;
            V.rowgrp = (const PgRowGrp*)(t + toff[l].rowgrp);
            V.tilex = (const int32_t*)(t + toff[l].tilex);
            V.pyrCpr = toff[l].cpr; V.pyrRows = toff[l].prows; V.pyrGpw = pyrGpw;
            V.qtab2 = toff[l].hasQ2 ? (const PgQuadTab2*)(t + toff[l].qtab2) : nullptr;
            if (toff[l].fnb > 0) {                          // the fused tables belong to the SOURCE level l - 1
                PgFuseLevel& SV = c->fuse.lvl[l - 1];
                SV.bands = (const int32_t*)(t + toff[l].fbands); SV.cols = (const int32_t*)(t + toff[l].fcols);
                SV.nBands = toff[l].fnb; SV.spb = toff[l].fspb; SV.rows = toff[l].frows;
            }
        }
    }
    P.cellCand = (uint32_t*)c->cellCand.p; P.cellCount = (int32_t*)c->cellCount.p;
    {
        std::vector<uint32_t> ct((size_t)cells * 16, 0u);
        for (int l = 0; l < L; l++) {
            const PgLevel& V = P.lvl[l];
            const int maxBorderX = V.w - PG_EDGE, maxBorderY = V.h - PG_EDGE;
            for (int i = 0; i < V.nRows; i++)
                for (int j = 0; j < V.nCols; j++) {
                    const int cidx = i * V.nCols + j;
                    uint32_t* r = &ct[((size_t)V.cellBase + cidx) * 16];
                    const int iniY = PG_EDGE + i * V.hCell, iniX = PG_EDGE + j * V.wCell;      // :791-801
                    const int W = std::min(iniX + V.wCell + 6, maxBorderX) - iniX;
                    const int H = std::min(iniY + V.hCell + 6, maxBorderY) - iniY;
                    // skipped cells (:794, :803) and windows cv::FAST finds nothing in (< 7 px)
                    const bool skip = iniY >= maxBorderY - 3 || iniX >= maxBorderX - 6 || W < 7 || H < 7;
                    const uint64_t off = (uint64_t)(pyrOff[l] * B) + (uint64_t)iniY * V.pitch + (uint64_t)(iniX - 1);
                    r[0] = (uint32_t)l | ((uint32_t)(V.cellBase + cidx) << 4);
                    r[1] = (uint32_t)iniX | ((uint32_t)iniY << 16);
                    r[2] = (uint32_t)(skip ? 0 : W) | ((uint32_t)(skip ? 0 : H) << 8) | ((uint32_t)skip << 16) | ((uint32_t)V.cellCap << 17);
                    r[3] = (uint32_t)V.pitch;
                    r[4] = (uint32_t)off; r[5] = (uint32_t)(off >> 32);
                    r[6] = (uint32_t)V.fstride;
                    r[7] = (uint32_t)(V.cellCandOff + (int64_t)cidx * V.cellCap);
                    // w8-w15: which lanes / result bits of the necessary test lie inside the interior, as LANE MASKS and bit
                    // patterns (fast.hip, quick_pass_b: lane = quad (lane & 7) x row lr = 2 ((lane >> 3) & 3) + (lane >> 5), steps of 8
                    // rows) -- a per-cell constant that used to cost every wave 17 vector instructions
                    const int IW = W - 6, IH = H - 6;
                    if (!skip && IW <= 32 && IH <= 40) {
                        const int qFull = IW >> 2, rem = IW & 3, base = IH >> 3, rr = IH & 7;
                        r[8] = ((1u << qFull) - 1u) * 0x01010101u;                     // lanes of whole quads (both 32-lane halves)
                        r[9] = qFull < 8 ? (1u << qFull) * 0x01010101u : 0u;           // lanes of the partial quad
                        r[10] = (1u << (8 * rem)) - 1u;                                // its pixels, one byte each
                        const int kLo = (rr + 1) >> 1, kHi = rr >> 1;                  // lanes whose row takes one step more than IH / 8
                        r[11] = kLo >= 4 ? 0xFFFFFFFFu : (1u << (8 * kLo)) - 1u;       //   lanes 0-31: lr = 2 (lane >> 3)
                        r[12] = (1u << (8 * kHi)) - 1u;                                //   lanes 32-63: lr = 2 ((lane >> 3) & 3) + 1
                        r[13] = ((1u << std::min(base, 4)) - 1u) * 0x11111111u;        // step bits every row has
                        r[14] = base < 4 ? 0x11111111u << base : 0u;                   // the step more
                        r[15] = base >= 5 ? 2u : (base == 4 ? 1u : 0u);                // fifth step: all rows / the rows of w11-w12 / none
                    }
                }
        }
        // + slack: a K2 wave loads its record BEFORE it knows whether the position exists (fast.hip), and the last workgroup's positions
        // run up to waves per workgroup x records per wave (4 x 64) past the table (round 6: 8 records of slack faulted under
        // "fast_cells_per_wave" = 64 when the table happened to end a mapping)
        const size_t cellTabSlack = (size_t)(4 * 64 + 8) * 64;
        if ((rc = ensure(c, c->cellTab, ct.size() * 4 + cellTabSlack))) return rc;
        PG_HIP(c, hipMemcpy(c->cellTab.p, ct.data(), ct.size() * 4, hipMemcpyHostToDevice));
        P.cellTab = (const uint32_t*)c->cellTab.p;
        P.pyrBase = (const uint8_t*)c->pyr.p;
        // balanced dispatch order (fast.hip, k_fast_cells): XCD x takes cells [n_l x / 8, n_l (x + 1) / 8) of every level l
        std::vector<std::vector<int>> lists(8);
        for (int x = 0; x < 8; x++)
            for (int l = 0; l < L; l++) {
                const PgLevel& V = P.lvl[l];
                const int n = V.nCols * V.nRows;
                for (int k = (int)((int64_t)n * x / 8); k < (int)((int64_t)n * (x + 1) / 8); k++) lists[x].push_back(V.cellBase + k);
            }
        size_t per = 0;
        for (auto& v : lists) per = std::max(per, v.size());
        std::vector<uint32_t> cb(8 * per * 16, 0u);
        for (int x = 0; x < 8; x++)
            for (size_t k = 0; k < per; k++) {
                uint32_t* r = &cb[((size_t)x * per + k) * 16];
                if (k < lists[x].size()) memcpy(r, &ct[(size_t)lists[x][k] * 16], 64);
                else { r[0] = 0xFFFFFFF0u; r[2] = 1u << 16; }                                // padding: the wave returns at once
            }
        if ((rc = ensure(c, c->cellTabBal, cb.size() * 4 + cellTabSlack))) return rc;
        PG_HIP(c, hipMemcpy(c->cellTabBal.p, cb.data(), cb.size() * 4, hipMemcpyHostToDevice));
        P.cellTabBal = (const uint32_t*)c->cellTabBal.p;
        P.cellsPerXcdBal = (int)per;
    }
    {
        // K3's coordinate tables for its pass kernel (quadtree.hip, k_qt_leaves: leaf column / row and candidate-order rank are separable
        // in x and y), per level: [regionW] x entries, [regionH] y entries, [2^D + 1] first y of every leaf row
        std::vector<uint2> qt;
        for (int l = 0; l < L; l++) {
            PgLevel& V = P.lvl[l];
            const int regionW = V.w - 2 * PG_EDGE, regionH = V.h - 2 * PG_EDGE, D = qt_pyr_depth(V.nIni);
            V.qtTabOff = (int32_t)qt.size();
            for (int x = 0; x < regionW; x++) {
                const uint2 e = qt_tab_entry(false, x, V.hX, V.nIni, regionH, D, V.wCell, V.hCell, V.nCols);
                qt.push_back(make_uint2(((e.x >> (2 * D)) << D) | qt_compact_bits(e.x & ((1u << (2 * D)) - 1)), e.y));
            }
            std::vector<int> rowY((size_t)(1 << D) + 1, regionH);
            for (int y = regionH - 1; y >= 0; y--) {
                const uint2 e = qt_tab_entry(true, y, V.hX, V.nIni, regionH, D, V.wCell, V.hCell, V.nCols);
                const uint32_t row = qt_compact_bits(e.x >> 1);
                for (uint32_t r = 0; r <= row; r++) rowY[r] = std::min(rowY[r], y);       // first y whose row is >= r
            }
            for (int y = 0; y < regionH; y++) {
                const uint2 e = qt_tab_entry(true, y, V.hX, V.nIni, regionH, D, V.wCell, V.hCell, V.nCols);
                qt.push_back(make_uint2(qt_compact_bits(e.x >> 1), e.y));
            }
            for (int r = 0; r <= (1 << D); r++) qt.push_back(make_uint2((uint32_t)rowY[r], 0u));
        }
        if ((rc = ensure(c, c->qtTab, qt.size() * sizeof(uint2) + 64))) return rc;
        PG_HIP(c, hipMemcpy(c->qtTab.p, qt.data(), qt.size() * sizeof(uint2), hipMemcpyHostToDevice));
        const size_t leafBytes = (size_t)B * L * PG_QT_LEAF_CAP * sizeof(uint2);
        if ((rc = ensure(c, c->qtLeaf, leafBytes))) return rc;
        P.qtTab = (const uint2*)c->qtTab.p; P.qtLeaf = (uint2*)c->qtLeaf.p; P.qtSplit = c->qtSplit; P.qtThreads = c->qtThreads; P.qtWide = getenv("PGORB_QT_WIDE") ? atoi(getenv("PGORB_QT_WIDE")) != 0 : 1;
    }
    P.fastTilePitch = c->fastTilePitch; P.fastWpb = c->fastWpb; P.fastCpw = c->fastCpw; c->fuse.enabled = c->fused;
    P.cand = (uint32_t*)c->cand.p; P.sel = (uint32_t*)c->sel.p;
    P.nodeScratch = (int32_t*)c->nodes.p;
    P.candCount = (int32_t*)c->counters.p;
    P.kpCount = P.candCount + (size_t)B * PG_MAXL;
    // the status word heads the result block of the host-frame calls (pgorb_extract*: one download brings status, counts,
    // keypoints and descriptors); K1's first launch of a batch clears it
    {
        const size_t ob = 64 + (((size_t)B * 4 + 63) & ~(size_t)63) + (((size_t)B * selTotal * sizeof(pgorb_keypoint) + 63) & ~(size_t)63) + (size_t)B * selTotal * 32 + 64;
        if ((rc = ensure(c, c->outBlk, ob))) return rc;
        PG_HIP(c, hipMemset(c->outBlk.p, 0, 64));
        // the memset runs on the null stream and may return before it has executed; the host-frame calls launch on a private
        // NON-BLOCKING stream (no implicit order with stream 0), so make it land before anything can write a status
        PG_HIP(c, hipStreamSynchronize(nullptr));
    }
    P.status = (int32_t*)c->outBlk.p;
    c->planW = w; c->planH = h; c->planBatch = B; c->planValid = true;
    c->planEpoch++;                                            // (captured graphs hold the old plan by value)
    return 0;
}

// PGORB_DEBUG_SYNC=1: wait behind every kernel of the one-launch-per-kernel path and say which one completed (a GPU memory fault
// then names its kernel: the last line printed is the kernel BEFORE the faulting one)
#define PG_DBG_SYNC(name) do { static const bool dbg_ = getenv("PGORB_DEBUG_SYNC") != nullptr; \
                               if (dbg_) { (void)hipStreamSynchronize(s); fprintf(stderr, "[pgorb] %s done (%dx%d x %d)\n", name, w, h, nframes); fflush(stderr); } } while (0)
int run_batch(pgorb_ctx* c, const uint8_t* d_gray, bool resident_in_level0, int nframes, int w, int h,
              int stride, int64_t frame_stride, pgorb_keypoint* d_kps, uint8_t* d_desc,
              int cap_per_frame, int32_t* d_n, hipStream_t s)
{
    PgPlan P = c->plan;                                   // by-value copy handed to the kernels
    c->lastAliased = false;
    if (!resident_in_level0) {
        const bool aligned = ((uintptr_t)d_gray % 4 == 0) && (stride % 4 == 0) && (frame_stride % 4 == 0);
        if (aligned) {                                    // zero-copy: level 0 is the caller's buffer
            P.lvl[0].img = const_cast<uint8_t*>(d_gray);
            P.lvl[0].pitch = stride;
            P.lvl[0].fstride = frame_stride;
            c->lastAliased = true;
        } else {
            pg_launch_copy_level0(P, d_gray, stride, frame_stride, nframes, s);
        }
    }
    // the device status word is per batch: cleared by the first pyramid launch (pyramid.hip) where the order of the kernels allows it,
    // by a memset otherwise
    const bool fusedPath = c->fuse.enabled && !c->pipePyr && !c->pipeLev && P.nlevels > 1;
    const bool foldClear = !c->pipePyr && P.nlevels > 1 && !fusedPath;
    if (!foldClear) PG_HIP(c, hipMemsetAsync(P.status, 0, 16, s));
    hipEvent_t* ev = (c->profExtract < c->profMax) ? &c->evExtract[5 * (size_t)c->profExtract] : nullptr;
    if (ev) PG_HIP(c, hipEventRecord(ev[0], s));
    if (c->pipePyr && P.nlevels > 1) {
        // K1 is HBM-bound and K2 VALU-issue-bound, and K2 of level l only needs level l: the resize chain runs on a
        // high-priority side stream, K2 level by level on another, each level's K2 behind the launch that wrote it.
        // (Stage events: "pyramid" = start .. end of the chain, "fast" = end of the chain .. end of K2: they overlap.)
        if (!c->sPyr) {
            int lo = 0, hi = 0;
            PG_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
            PG_HIP(c, hipStreamCreateWithPriority(&c->sPyr, hipStreamNonBlocking, hi));
            PG_HIP(c, hipStreamCreateWithPriority(&c->sFast, hipStreamNonBlocking, lo));
            PG_HIP(c, hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
            PG_HIP(c, hipEventCreateWithFlags(&c->evPyrDone, hipEventDisableTiming));
            PG_HIP(c, hipEventCreateWithFlags(&c->evFastDone, hipEventDisableTiming));
            for (int l = 0; l < PG_MAXL; l++) PG_HIP(c, hipEventCreateWithFlags(&c->evLevel[l], hipEventDisableTiming));
        }
        PG_HIP(c, hipEventRecord(c->evFork, s));
        PG_HIP(c, hipStreamWaitEvent(c->sPyr, c->evFork, 0));
        PG_HIP(c, hipStreamWaitEvent(c->sFast, c->evFork, 0));
        pg_launch_fast_levels(P, nframes, 0, 1, c->sFast);
        for (int l = 1; l < P.nlevels; l++) {
            pg_launch_pyramid_level(P, l, nframes, c->sPyr);
            PG_HIP(c, hipEventRecord(c->evLevel[l], c->sPyr));
            PG_HIP(c, hipStreamWaitEvent(c->sFast, c->evLevel[l], 0));
            pg_launch_fast_levels(P, nframes, l, l + 1, c->sFast);
        }
        PG_HIP(c, hipEventRecord(c->evPyrDone, c->sPyr));
        PG_HIP(c, hipEventRecord(c->evFastDone, c->sFast));
        PG_HIP(c, hipStreamWaitEvent(s, c->evPyrDone, 0));
        if (ev) PG_HIP(c, hipEventRecord(ev[1], s));
        PG_HIP(c, hipStreamWaitEvent(s, c->evFastDone, 0));
        if (ev) PG_HIP(c, hipEventRecord(ev[2], s));
    } else if (c->pipeLev && P.nlevels > 1) {
        // K2 group by group on the caller's stream; K3 of a group on a second stream as soon as its K2 is done, K4-6 of a
        // group on a third as soon as its K3 is done: the latency-bound quadtree of one group runs under the issue-bound
        // kernels of the others.  (Stage events: "fast" = K2 of all groups, "quadtree" = the wait for the side streams.)
        if (!c->sQt) {
            int lo = 0, hi = 0;
            PG_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
            PG_HIP(c, hipStreamCreateWithPriority(&c->sQt, hipStreamNonBlocking, c->pipeLevPrio ? hi : lo));
            PG_HIP(c, hipStreamCreateWithPriority(&c->sDesc, hipStreamNonBlocking, lo));
            PG_HIP(c, hipEventCreateWithFlags(&c->evDescDone, hipEventDisableTiming));
            for (int l = 0; l < PG_MAXL; l++) {
                PG_HIP(c, hipEventCreateWithFlags(&c->evGrpFast[l], hipEventDisableTiming));
                PG_HIP(c, hipEventCreateWithFlags(&c->evGrpQt[l], hipEventDisableTiming));
            }
        }
        for (int l = 1; l < P.nlevels; l++)
            if (!pg_launch_pyramid_level(P, l, nframes, s, l == 1 ? P.status : nullptr) && l == 1) PG_HIP(c, hipMemsetAsync(P.status, 0, 16, s));
        if (ev) PG_HIP(c, hipEventRecord(ev[1], s));
        for (int beg = 0; beg < P.nlevels;) {
            int end = beg + 1;
            while (end < P.nlevels && !((c->pipeLev >> end) & 1)) end++;
            pg_launch_fast_levels(P, nframes, beg, end, s);
            PG_HIP(c, hipEventRecord(c->evGrpFast[beg], s));
            PG_HIP(c, hipStreamWaitEvent(c->sQt, c->evGrpFast[beg], 0));
            pg_launch_quadtree_levels(P, nframes, beg, end, c->sQt);
            PG_HIP(c, hipEventRecord(c->evGrpQt[beg], c->sQt));
            PG_HIP(c, hipStreamWaitEvent(c->sDesc, c->evGrpQt[beg], 0));
            pg_launch_describe_levels(P, nframes, d_kps, d_desc, cap_per_frame, d_n, beg, end, c->sDesc);
            beg = end;
        }
        if (ev) PG_HIP(c, hipEventRecord(ev[2], s));
        PG_HIP(c, hipEventRecord(c->evDescDone, c->sDesc));
        PG_HIP(c, hipStreamWaitEvent(s, c->evDescDone, 0));
        if (ev) { PG_HIP(c, hipEventRecord(ev[3], s)); PG_HIP(c, hipEventRecord(ev[4], s)); c->profExtract++; }
        PG_HIP(c, hipGetLastError());
        c->lastFrames = nframes;
        return 0;
    } else if (c->fuse.enabled && P.nlevels > 1) {
        // Every level read ONCE (fused.hip): the launch that resizes level l -> l + 1 detects level l; a level without fused tables
        // takes K1 + its own K2; the last level is detected by K2.  (Stage events: "pyramid" = the chain of fused launches, i.e. the
        // whole pyramid AND the detection of levels 0 .. L-2; "fast" = what is left for K2.)
        PG_HIP(c, hipMemsetAsync(P.status, 0, 16, s));        // (K2's part of the first launch may report: the word is cleared in front of it)
        int pending = -1;                                     // first level of a run of levels still waiting for K2
        for (int l = 0; l + 1 < P.nlevels; l++) {
            if (pg_launch_pyr_fast(P, c->fuse, l, nframes, s)) {
                if (pending >= 0) { pg_launch_fast_levels(P, nframes, pending, l, s); pending = -1; }
            } else {
                pg_launch_pyramid_level(P, l + 1, nframes, s);
                if (pending < 0) pending = l;
            }
        }
        if (ev) PG_HIP(c, hipEventRecord(ev[1], s));
        if (c->evPyrEnd) PG_HIP(c, hipEventRecord(c->evPyrEnd, s));
        pg_launch_fast_levels(P, nframes, pending >= 0 ? pending : P.nlevels - 1, P.nlevels, s);
        if (ev) PG_HIP(c, hipEventRecord(ev[2], s));
        if (c->evFastEnd) PG_HIP(c, hipEventRecord(c->evFastEnd, s));
    } else {
        for (int l = 1; l < P.nlevels; l++) {
            if (!pg_launch_pyramid_level(P, l, nframes, s, l == 1 ? P.status : nullptr) && l == 1) PG_HIP(c, hipMemsetAsync(P.status, 0, 16, s));
            PG_DBG_SYNC("K1 pyramid level");
        }
        if (ev) PG_HIP(c, hipEventRecord(ev[1], s));
        if (c->evPyrEnd) PG_HIP(c, hipEventRecord(c->evPyrEnd, s));
        pg_launch_fast(P, nframes, s);
        PG_DBG_SYNC("K2 fast");
        if (ev) PG_HIP(c, hipEventRecord(ev[2], s));
        if (c->evFastEnd) PG_HIP(c, hipEventRecord(c->evFastEnd, s));
    }
    pg_launch_quadtree(P, nframes, s);
    PG_DBG_SYNC("K3 quadtree");
    if (ev) PG_HIP(c, hipEventRecord(ev[3], s));
    pg_launch_describe(P, nframes, d_kps, d_desc, cap_per_frame, d_n, s);
    PG_DBG_SYNC("K4-6 describe");
    if (ev) { PG_HIP(c, hipEventRecord(ev[4], s)); c->profExtract++; }
    PG_HIP(c, hipGetLastError());
    c->lastFrames = nframes;
    return 0;
}

}  // namespace

// ---- helpers used by bow.hip ---------------------------------------------------------------
extern "C" void pg_forward_option_to_lanes(pgorb_ctx* c, const char* key, int value);      // (defined behind pgorb_stream)
int pg_ctx_fail(pgorb_ctx* c, int code, const char* msg) { return fail(c, code, "%s", msg); }
int pg_ctx_device(pgorb_ctx* c) { return c->prm.device; }
int pg_ctx_stage(pgorb_ctx* c, int which, size_t bytes, void** p)
{
    Arena* a = which == 0 ? &c->stageA : which == 1 ? &c->stageB : which == 3 ? &c->stageSfi : &c->stageOut;
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc = ensure(c, *a, bytes);
    if (rc) return rc;
    *p = a->p;
    return 0;
}
// The matchers' scratch arena (lists, bins) for work queued on stream `s`: SearchForInitialization, the projection searches and
// SearchByBoW share ONE arena per context, and a caller may queue them on different streams -- the new user waits for the
// previous user's event (round-3 advisory: BoW's memset of the bins could overtake a projection call still reading its lists).
// pg_ctx_scratch_done records the event after the last launch that touches the arena.
int pg_ctx_scratch(pgorb_ctx* c, size_t bytes, hipStream_t s, void** p)
{
    int rc = pg_ctx_stage(c, 3, bytes, p);
    if (rc) return rc;
    if (c->sfiUsed && c->sfiStream != s) PG_HIP(c, hipStreamWaitEvent(s, c->evSfi, 0));
    return 0;
}
int pg_ctx_scratch_done(pgorb_ctx* c, hipStream_t s)
{
    if (!c->evSfi) PG_HIP(c, hipEventCreateWithFlags(&c->evSfi, hipEventDisableTiming));
    PG_HIP(c, hipEventRecord(c->evSfi, s));
    c->sfiStream = s; c->sfiUsed = true;
    return 0;
}
int pg_ctx_pinned(pgorb_ctx* c, size_t bytes, void** p)
{
    if (c->pinnedBytes < bytes) {
        PG_HIP(c, hipSetDevice(c->prm.device));
        if (c->pinned) (void)hipHostFree(c->pinned);
        c->pinned = nullptr; c->pinnedBytes = 0;
        const size_t want = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
        PG_HIP(c, hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
        c->pinnedBytes = want;
    }
    *p = c->pinned;
    return 0;
}
void pg_ctx_vocab_drop(pgorb_ctx* c);
// header checks of a vocabulary blob of `nbytes` bytes (bow.hip, blob layout): magic, version, and that the sections the
// header implies fit; the structure itself is checked by pgorb_vocab_from_blob / the loader on the host path and by
// k_vocab_validate on the device path
static int vocab_header_ok(pgorb_ctx* c, const int32_t* hdr, size_t nbytes)
{
    if (hdr[0] != 0x43564750 || hdr[1] != 1 || hdr[4] < 2)
        return fail(c, PGORB_E_ARG, "not a pgorb vocabulary blob");
    const size_t n = (size_t)hdr[4];
    auto pad = [](size_t v) { return (v + 63) / 64 * 64; };
    size_t need = 64;
    need = pad(need + n * 32); need = pad(need + n * 8);
    for (int k = 0; k < 4; k++) need = pad(need + n * 4);
    need = pad(need + (n - 1) * 4);
    if (nbytes < need)
        return fail(c, PGORB_E_ARG, "vocabulary blob truncated: %zu bytes, header implies %zu", nbytes, need);
    return 0;
}
int pg_ctx_vocab_store(pgorb_ctx* c, const void* src, size_t nbytes, bool src_on_device, hipStream_t s)
{
    PG_HIP(c, hipSetDevice(c->prm.device));
    int32_t hdr[16];
    if (src_on_device) {
        PG_HIP(c, hipMemcpyAsync(hdr, src, 64, hipMemcpyDeviceToHost, s));
        PG_HIP(c, hipStreamSynchronize(s));
    } else memcpy(hdr, src, 64);
    int rc = vocab_header_ok(c, hdr, nbytes);
    if (rc) return rc;
    if ((rc = ensure(c, c->vocab, nbytes))) return rc;
    if (src_on_device) PG_HIP(c, hipMemcpyAsync(c->vocab.p, src, nbytes, hipMemcpyDeviceToDevice, s));
    else PG_HIP(c, hipMemcpy(c->vocab.p, src, nbytes, hipMemcpyHostToDevice));
    c->vocabK = hdr[2]; c->vocabL = hdr[3]; c->vocabNodes = hdr[4];
    return 0;
}
// The receive side of the vocabulary broadcast (comm.hip): room for `nbytes` in the context's vocabulary arena (the collective
// writes there directly, no second copy), then -- once the bytes have landed on stream `s` -- the header checks and the fields.
int pg_ctx_vocab_reserve(pgorb_ctx* c, size_t nbytes, void** p)
{
    PG_HIP(c, hipSetDevice(c->prm.device));
    pg_ctx_vocab_drop(c);
    int rc = ensure(c, c->vocab, nbytes);
    if (rc) return rc;
    *p = c->vocab.p;
    return 0;
}
int pg_ctx_vocab_commit(pgorb_ctx* c, size_t nbytes, hipStream_t s)
{
    PG_HIP(c, hipSetDevice(c->prm.device));
    int32_t hdr[16];
    PG_HIP(c, hipMemcpyAsync(hdr, c->vocab.p, 64, hipMemcpyDeviceToHost, s));
    PG_HIP(c, hipStreamSynchronize(s));
    int rc = vocab_header_ok(c, hdr, nbytes);
    if (rc) return rc;
    c->vocabK = hdr[2]; c->vocabL = hdr[3]; c->vocabNodes = hdr[4];
    return 0;
}
void pg_ctx_vocab_drop(pgorb_ctx* c) { c->vocabK = c->vocabL = c->vocabNodes = 0; }
int pg_ctx_vocab_get(pgorb_ctx* c, const uint8_t** d_blob, int* k, int* L, int* nnodes)
{
    if (!c->vocab.p || !c->vocabNodes) return fail(c, PGORB_E_ARG, "no vocabulary uploaded");
    *d_blob = (const uint8_t*)c->vocab.p; *k = c->vocabK; *L = c->vocabL; *nnodes = c->vocabNodes;
    return 0;
}

extern "C" {

int pgorb_create(const pgorb_params* p, pgorb_ctx** out)
{
    if (!p || !out) return fail(nullptr, PGORB_E_ARG, "null argument");
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > PG_MAXL || p->nfeatures < 1 || !(p->scale_factor > 1.0f) ||
        p->max_batch < 1 || p->max_width < 62 || p->max_height < 62 || p->min_th_fast < 1 ||
        p->ini_th_fast < p->min_th_fast || p->ini_th_fast > 254)
        return fail(nullptr, PGORB_E_ARG, "invalid parameters");
    if (p->max_width > 4095 || p->max_height > 4095)
        return fail(nullptr, PGORB_E_LIMIT, "max_width/max_height above 4095 not supported");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev)
        return fail(nullptr, PGORB_E_NODEVICE, "no HIP device %d (found %d); libpgorb has no CPU path",
                    p->device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) != hipSuccess)
        return fail(nullptr, PGORB_E_NODEVICE, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PGORB_E_NODEVICE, "device %d is %s; libpgorb is built for gfx950 only",
                    p->device, prop.gcnArchName);
    if (hipSetDevice(p->device) != hipSuccess)
        return fail(nullptr, PGORB_E_HIP, "hipSetDevice failed");

    pgorb_ctx* c = new pgorb_ctx();
    c->prm = *p;
    c->mx = pg_match_default_opts();
    if (const char* e = getenv("PGORB_EXTRACT_CHUNK_KB")) { const int kb = atoi(e); if (kb >= 16) c->chunkBytes = kb << 10; }
    c->noStage = getenv("PGORB_EXTRACT_STAGE") == nullptr;
    c->useGraph = getenv("PGORB_EXTRACT_NO_GRAPH") == nullptr;
    if (const char* e = getenv("PGORB_QT_THREADS")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) c->qtThreads = v; }
    if (const char* e = getenv("PGORB_QT_SPLIT")) { const int v = atoi(e); if (v >= 0 && v <= 2) c->qtSplit = v; }      // A / B switch of K3's two forms (option "quadtree_split")
    // ORBextractor.cc:415-446
    const int L = p->nlevels;
    c->scaleFactor = p->scale_factor;
    c->mvScaleFactor[0] = 1.0f; c->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i <= L; i++) {
        c->mvScaleFactor[i] = (float)(c->mvScaleFactor[i - 1] * c->scaleFactor);
        c->mvLevelSigma2[i] = c->mvScaleFactor[i] * c->mvScaleFactor[i];
    }
    for (int i = 0; i <= L; i++) {
        c->mvInvScaleFactor[i] = 1.0f / c->mvScaleFactor[i];
        c->mvInvLevelSigma2[i] = 1.0f / c->mvLevelSigma2[i];
    }
    const float factor = (float)(1.0f / c->scaleFactor);
    float nDesired = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L; l++) {
        c->mnFeaturesPerLevel[l] = cvRound(nDesired);
        sum += c->mnFeaturesPerLevel[l];
        nDesired *= factor;
    }
    c->mnFeaturesPerLevel[L] = std::max(p->nfeatures - sum, 0);
    *out = c;
    return 0;
}

void pgorb_destroy(pgorb_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->prm.device);
    while (!c->streams.empty()) pgorb_stream_destroy(c->streams.back());      // a stream holds a pointer to its context
    if (c->evPyrEnd) (void)hipEventDestroy(c->evPyrEnd);
    if (c->evFastEnd) (void)hipEventDestroy(c->evFastEnd);
    Arena* all[] = {&c->cellTab, &c->cellTabBal, &c->cellCand, &c->cellCount, &c->pyr, &c->cand, &c->sel, &c->nodes, &c->counters, &c->tables, &c->qtTab, &c->qtLeaf,
                    &c->outBlk, &c->stageA, &c->stageB, &c->stageOut, &c->stageSfi, &c->vocab, &c->xdesc};
    for (Arena* a : all) if (a->p) (void)hipFree(a->p);
    if (c->sPyr) {
        (void)hipStreamSynchronize(c->sPyr); (void)hipStreamSynchronize(c->sFast);
        (void)hipStreamDestroy(c->sPyr); (void)hipStreamDestroy(c->sFast);
        (void)hipEventDestroy(c->evFork); (void)hipEventDestroy(c->evPyrDone); (void)hipEventDestroy(c->evFastDone);
        for (int l = 0; l < PG_MAXL; l++) (void)hipEventDestroy(c->evLevel[l]);
    }
    if (c->sQt) {
        (void)hipStreamSynchronize(c->sQt); (void)hipStreamSynchronize(c->sDesc);
        (void)hipStreamDestroy(c->sQt); (void)hipStreamDestroy(c->sDesc);
        (void)hipEventDestroy(c->evDescDone);
        for (int l = 0; l < PG_MAXL; l++) { (void)hipEventDestroy(c->evGrpFast[l]); (void)hipEventDestroy(c->evGrpQt[l]); }
    }
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinIn) (void)hipHostFree(c->pinIn);
    if (c->hg.exec) (void)hipGraphExecDestroy(c->hg.exec);
    if (c->hg.g) (void)hipGraphDestroy(c->hg.g);
    if (c->sHost) { (void)hipStreamSynchronize(c->sHost); (void)hipStreamDestroy(c->sHost); }
    if (c->evSfi) (void)hipEventDestroy(c->evSfi);
    for (hipEvent_t e : c->evExtract) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->evMatch) (void)hipEventDestroy(e);
    delete c;
}

const char* pgorb_last_error(const pgorb_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int pgorb_levels(const pgorb_ctx* c) { return c ? c->prm.nlevels : PGORB_E_ARG; }

int pgorb_scale_tables(const pgorb_ctx* c, float* scale, float* inv, float* s2, float* is2)
{
    if (!c) return PGORB_E_ARG;
    const size_t n = (size_t)(c->prm.nlevels + 1) * sizeof(float);
    if (scale) memcpy(scale, c->mvScaleFactor, n);
    if (inv) memcpy(inv, c->mvInvScaleFactor, n);
    if (s2) memcpy(s2, c->mvLevelSigma2, n);
    if (is2) memcpy(is2, c->mvInvLevelSigma2, n);
    return 0;
}

int pgorb_features_per_level(const pgorb_ctx* c, int32_t* out)
{
    if (!c || !out) return PGORB_E_ARG;
    for (int i = 0; i <= c->prm.nlevels; i++) out[i] = c->mnFeaturesPerLevel[i];
    return 0;
}

int pgorb_max_keypoints(const pgorb_ctx* c, int w, int h)
{
    if (!c) return PGORB_E_ARG;
    if (w > c->prm.max_width || h > c->prm.max_height) return PGORB_E_LIMIT;
    LevelGeom g[PG_MAXL];
    int rc = level_geometry(c, w, h, g);
    if (rc) return rc;
    int total = 0;
    for (int l = 0; l < c->prm.nlevels; l++) total += std::max(c->mnFeaturesPerLevel[l] + 2, 4 * g[l].nIni);
    return total;
}

int pgorb_extract_batch_device(pgorb_ctx* c, const uint8_t* d_gray, int nframes, int w, int h,
                               int stride, int64_t frame_stride, pgorb_keypoint* d_kps,
                               uint8_t* d_desc, int cap_per_frame, int32_t* d_n, void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!d_gray || !d_kps || !d_desc || !d_n || nframes < 1 || w < 1 || h < 1 || stride < w ||
        cap_per_frame < 1)
        return fail(c, PGORB_E_ARG, "bad argument to pgorb_extract_batch_device");
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc = make_plan(c, w, h, nframes);
    if (rc) return rc;
    return run_batch(c, d_gray, false, nframes, w, h, stride, frame_stride, d_kps, d_desc,
                     cap_per_frame, d_n, (hipStream_t)stream);
}

int pgorb_extract_batch_color_device(pgorb_ctx* c, const uint8_t* d_img, int nframes, int w, int h, int stride,
                                     int64_t frame_stride, int channels, int rgb_order, pgorb_keypoint* d_kps,
                                     uint8_t* d_desc, int cap_per_frame, int32_t* d_n, void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!d_img || !d_kps || !d_desc || !d_n || nframes < 1 || w < 1 || h < 1 || (channels != 3 && channels != 4) ||
        stride < w * channels || cap_per_frame < 1)
        return fail(c, PGORB_E_ARG, "bad argument to pgorb_extract_batch_color_device");
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc = make_plan(c, w, h, nframes);
    if (rc) return rc;
    pg_launch_color_to_gray(c->plan, d_img, stride, frame_stride, channels, rgb_order, nframes, (hipStream_t)stream);
    return run_batch(c, nullptr, true, nframes, w, h, w, 0, d_kps, d_desc, cap_per_frame, d_n, (hipStream_t)stream);
}

int pgorb_extract_batch_ingest_device(pgorb_ctx* c, const uint8_t* d_img, int nframes, int src_w, int src_h, int stride,
                                      int64_t frame_stride, int channels, int rgb_order, int rotate_degrees,
                                      int vertical_flip, int horizontal_flip, pgorb_keypoint* d_kps, uint8_t* d_desc,
                                      int cap_per_frame, int32_t* d_n, void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!d_img || !d_kps || !d_desc || !d_n || nframes < 1 || src_w < 1 || src_h < 1 ||
        (channels != 1 && channels != 3 && channels != 4) || stride < src_w * channels || cap_per_frame < 1)
        return fail(c, PGORB_E_ARG, "bad argument to pgorb_extract_batch_ingest_device");
    if (rotate_degrees != 0 && rotate_degrees != 90 && rotate_degrees != 180 && rotate_degrees != 270)
        return fail(c, PGORB_E_ARG, "unsupported rotation %d: only multiples of 90 degrees", rotate_degrees);   // reader :203-207
    const bool swap = rotate_degrees == 90 || rotate_degrees == 270;
    const int w = swap ? src_h : src_w, h = swap ? src_w : src_h;
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc = make_plan(c, w, h, nframes);
    if (rc) return rc;
    pg_launch_ingest(c->plan, d_img, stride, frame_stride, src_w, src_h, channels, rgb_order, rotate_degrees / 90,
                     vertical_flip != 0, horizontal_flip != 0, nframes, (hipStream_t)stream);
    return run_batch(c, nullptr, true, nframes, w, h, w, 0, d_kps, d_desc, cap_per_frame, d_n, (hipStream_t)stream);
}

int pgorb_check_async(pgorb_ctx* c, void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!c->planValid) return 0;
    PG_HIP(c, hipSetDevice(c->prm.device));
    PG_HIP(c, hipStreamSynchronize((hipStream_t)stream));
    int32_t st = 0;
    PG_HIP(c, hipMemcpy(&st, c->plan.status, 4, hipMemcpyDeviceToHost));
    if (st) return fail(c, st, st == PGORB_E_TOOSMALL ? "device status %d (a pyramid level more than twice as tall as wide has candidates: the reference divides by zero there)"
                                                     : "device status %d (internal candidate capacity exceeded)", st);
    return 0;
}

// Frames in host memory, results to host memory: the reference's call shape (ORBextractor::operator() on a pageable cv::Mat,
// one frame per synchronous call: Frame.cc:251-257, Tracking.cc:262-266).  Round 4 cut the call's fixed costs:
//   * input: hipMemcpy2DAsync straight from the caller's (pageable) memory.  A page-locked staging buffer of the context's own,
//     filled in row chunks while the DMA engine moves the previous chunk, measured 24 us SLOWER per 1080p frame than the
//     runtime's pageable path (each extra copy command costs more than the overlap saves; PGORB_EXTRACT_STAGE=1 keeps the form);
//   * output: status word, counts, keypoints and descriptors live in ONE device block (PgPlan::status points into it) and come
//     back with one download and one synchronisation (there were two synchronous 4-byte copies in front of it).
// Host phases of the last calls: pgorb_profile_host.
int pgorb_extract_batch(pgorb_ctx* c, const uint8_t* const* gray, int nframes, int w, int h,
                        int stride, pgorb_keypoint* kps, uint8_t* desc, int cap, int* n)
{
    if (!c) return PGORB_E_ARG;
    if (!n) return fail(c, PGORB_E_ARG, "null count pointer");
    for (int f = 0; f < nframes; f++) n[f] = 0;
    if (nframes < 1 || !gray) return fail(c, PGORB_E_ARG, "bad argument to pgorb_extract_batch");
    if (w <= 0 || h <= 0) return 0;          // empty image: the reference returns silently (:1045)
    if (!kps || !desc || cap < 1 || stride < w) return fail(c, PGORB_E_ARG, "bad argument to pgorb_extract_batch");
    const auto t0 = std::chrono::steady_clock::now();
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc = make_plan(c, w, h, nframes);
    if (rc) return rc;
    for (int f = 0; f < nframes; f++) if (!gray[f]) return fail(c, PGORB_E_ARG, "null frame %d", f);
    const PgLevel& L0 = c->plan.lvl[0];
    const int need = c->plan.selTotal;       // the device block always holds the full bound
    // device result block of this call: status (64 B) | n[nframes] | kps[nframes][need] | desc[nframes][need][32]
    auto al64 = [](size_t v) { return (v + 63) & ~(size_t)63; };
    const size_t oN = 64, oK = oN + al64((size_t)nframes * 4), oD = oK + al64((size_t)nframes * need * sizeof(pgorb_keypoint)),
                 outBytes = oD + (size_t)nframes * need * 32;
    uint8_t* blk = (uint8_t*)c->outBlk.p;    // (make_plan sized it for max_batch frames)
    void* hv;
    if ((rc = pg_ctx_pinned(c, outBytes, &hv))) return rc;
    if (!c->sHost) PG_HIP(c, hipStreamCreateWithFlags(&c->sHost, hipStreamNonBlocking));
    hipStream_t hs = c->sHost;
    // ---- upload ----
    const size_t frameBytes = (size_t)w * h;
    const bool stage = frameBytes * nframes <= ((size_t)64 << 20) && !c->noStage;
    if (stage) {
        if (c->pinInBytes < frameBytes * nframes) {
            // an earlier call's chunk uploads out of the old buffer are complete (every call ends synchronised), but a caller
            // that got an error return in between may have left copies queued: drain before the pages go away
            if (c->pinIn) { (void)hipStreamSynchronize(hs); (void)hipHostFree(c->pinIn); }
            c->pinIn = nullptr; c->pinInBytes = 0;
            const size_t want = (frameBytes * std::min(nframes > 1 ? c->prm.max_batch : 1, (int)(((size_t)64 << 20) / frameBytes + 1)) + 4095) & ~(size_t)4095;
            PG_HIP(c, hipHostMalloc((void**)&c->pinIn, std::max(want, frameBytes * nframes), hipHostMallocDefault));
            c->pinInBytes = std::max(want, frameBytes * nframes);
        }
    }
    const int rowsPer = std::max(1, (int)(((size_t)c->chunkBytes + w - 1) / w));
    for (int f = 0; f < nframes; f++) {
        bool direct = !stage;
        if (stage) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, gray[f]) == hipSuccess && at.type == hipMemoryTypeHost) direct = true;   // already page-locked
            else (void)hipGetLastError();
        }
        uint8_t* dst = L0.img + (int64_t)f * L0.fstride;
        if (direct) {
            PG_HIP(c, hipMemcpy2DAsync(dst, L0.pitch, gray[f], stride, w, h, hipMemcpyHostToDevice, hs));
            continue;
        }
        uint8_t* pin = c->pinIn + (size_t)f * frameBytes;
        for (int r0 = 0; r0 < h; r0 += rowsPer) {
            const int rows = std::min(rowsPer, h - r0);
            if (stride == w) memcpy(pin + (size_t)r0 * w, gray[f] + (size_t)r0 * stride, (size_t)rows * w);
            else for (int r = 0; r < rows; r++) memcpy(pin + (size_t)(r0 + r) * w, gray[f] + (size_t)(r0 + r) * stride, w);
            if (L0.pitch == w) PG_HIP(c, hipMemcpyAsync(dst + (size_t)r0 * w, pin + (size_t)r0 * w, (size_t)rows * w, hipMemcpyHostToDevice, hs));
            else PG_HIP(c, hipMemcpy2DAsync(dst + (size_t)r0 * L0.pitch, L0.pitch, pin + (size_t)r0 * w, w, w, rows, hipMemcpyHostToDevice, hs));
        }
    }
    const auto t1 = std::chrono::steady_clock::now();
    // ---- kernels ----
    // Direct launches the first time a (plan, batch size) is seen -- the launchers may still allocate or configure --, captured
    // into a graph the second time, replayed from then on: the 10 launches + the download become one submission (the 7 resize
    // launches of a single frame are launch-bound: ~5 us apiece for ~2 us of work).  Not while a profile is armed (its events
    // would be captured) or a multi-stream pipeline option is on.
    pgorb_ctx::HostGraph& G = c->hg;
    const bool graphable = c->useGraph && c->profExtract >= c->profMax && !c->pipePyr && !c->pipeLev;
    const bool replay = graphable && G.exec && G.nframes == nframes && G.epoch == c->planEpoch && G.pinned == hv && G.outBytes == outBytes;
    if (replay) {
        PG_HIP(c, hipGraphLaunch(G.exec, hs));
        c->lastFrames = nframes; c->lastAliased = false;
    } else {
        const bool capture = graphable && G.seenFrames == nframes && G.seenEpoch == c->planEpoch;
        G.seenFrames = nframes; G.seenEpoch = c->planEpoch;
        if (capture) PG_HIP(c, hipStreamBeginCapture(hs, hipStreamCaptureModeRelaxed));
        rc = run_batch(c, nullptr, true, nframes, w, h, stride, 0, (pgorb_keypoint*)(blk + oK), blk + oD, need, (int32_t*)(blk + oN), hs);
        hipError_t e1 = rc ? hipSuccess : hipMemcpyAsync(hv, blk, outBytes, hipMemcpyDeviceToHost, hs);
        if (capture) {
            hipGraph_t g = nullptr;
            const hipError_t e2 = hipStreamEndCapture(hs, &g);
            if (rc || e1 != hipSuccess || e2 != hipSuccess || !g) {
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                c->useGraph = 0;                               // capture failed: direct launches, now and from here on
                if (rc) return rc;
                if ((rc = run_batch(c, nullptr, true, nframes, w, h, stride, 0, (pgorb_keypoint*)(blk + oK), blk + oD, need, (int32_t*)(blk + oN), hs))) return rc;
                PG_HIP(c, hipMemcpyAsync(hv, blk, outBytes, hipMemcpyDeviceToHost, hs));
                goto launched;
            }
            if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
            if (G.g) { (void)hipGraphDestroy(G.g); G.g = nullptr; }
            hipGraphExec_t ex = nullptr;
            if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGraphDestroy(g); (void)hipGetLastError(); c->useGraph = 0;
                if ((rc = run_batch(c, nullptr, true, nframes, w, h, stride, 0, (pgorb_keypoint*)(blk + oK), blk + oD, need, (int32_t*)(blk + oN), hs))) return rc;
                PG_HIP(c, hipMemcpyAsync(hv, blk, outBytes, hipMemcpyDeviceToHost, hs));
                goto launched;
            }
            G.g = g; G.exec = ex; G.nframes = nframes; G.epoch = c->planEpoch; G.pinned = hv; G.outBytes = outBytes;
            PG_HIP(c, hipGraphLaunch(G.exec, hs));
        } else {
            if (rc) return rc;
            if (e1 != hipSuccess) return fail(c, PGORB_E_HIP, "hipMemcpyAsync D2H failed: %s", hipGetErrorString(e1));
        }
    }
launched:
    const auto t2 = std::chrono::steady_clock::now();
    PG_HIP(c, hipStreamSynchronize(hs));
    const auto t3 = std::chrono::steady_clock::now();
    // ---- results ----
    const uint8_t* hb = (const uint8_t*)hv;
    const int32_t st = *(const int32_t*)hb;
    if (st) return fail(c, st, st == PGORB_E_TOOSMALL ? "device status %d (a pyramid level more than twice as tall as wide has candidates: the reference divides by zero there)"
                                                     : "device status %d (internal candidate capacity exceeded)", st);
    const int32_t* cnt = (const int32_t*)(hb + oN);
    for (int f = 0; f < nframes; f++)
        if (cnt[f] > cap)
            return fail(c, PGORB_E_CAP, "frame %d has %d keypoints, capacity %d", f, cnt[f], cap);
    for (int f = 0; f < nframes; f++) {
        n[f] = cnt[f];
        if (!cnt[f]) continue;
        memcpy(kps + (size_t)f * cap, hb + oK + (size_t)f * need * sizeof(pgorb_keypoint), (size_t)cnt[f] * sizeof(pgorb_keypoint));
        memcpy(desc + (size_t)f * cap * 32, hb + oD + (size_t)f * need * 32, (size_t)cnt[f] * 32);
    }
    const auto t4 = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    c->hostUs[0] += us(t0, t1); c->hostUs[1] += us(t1, t2); c->hostUs[2] += us(t2, t3); c->hostUs[3] += us(t3, t4); c->hostCalls++;
    return 0;
}

// mean host-side phase times (microseconds) of the pgorb_extract / pgorb_extract_batch calls since the last reset:
// us[0] input staging + upload issue, us[1] kernel launches + download issue, us[2] wait for the GPU, us[3] results to the caller's buffers.
// Returns the number of calls the sums cover; reset != 0 clears them (us may be NULL then).
int pgorb_profile_host(pgorb_ctx* c, double* us, int reset)
{
    if (!c) return PGORB_E_ARG;
    const int k = c->hostCalls;
    if (us) for (int i = 0; i < 4; i++) us[i] = c->hostUs[i];
    if (reset) { for (double& v : c->hostUs) v = 0; c->hostCalls = 0; }
    return k;
}

void* pgorb_host_alloc(int64_t bytes)
{
    void* p = nullptr;
    if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void pgorb_host_free(void* p) { if (p) (void)hipHostFree(p); }

int pgorb_host_register(void* p, int64_t bytes)
{
    if (!p || bytes <= 0) return PGORB_E_ARG;
    if (hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return PGORB_E_HIP; }
    return PGORB_OK;
}

int pgorb_host_unregister(void* p)
{
    if (!p) return PGORB_E_ARG;
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return PGORB_E_HIP; }
    return PGORB_OK;
}

int pgorb_extract(pgorb_ctx* c, const uint8_t* gray, int w, int h, int stride, pgorb_keypoint* kps,
                  uint8_t* desc, int cap, int* n)
{
    const uint8_t* frames[1] = {gray};
    if (c && n && (!gray || w <= 0 || h <= 0)) { *n = 0; return 0; }       // :1045
    return pgorb_extract_batch(c, frames, 1, w, h, stride, kps, desc, cap, n);
}

int pgorb_descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    if (!a || !b) return PGORB_E_ARG;
    int dist = 0;
    for (int i = 0; i < 32; i += 8) {
        uint64_t x, y;
        memcpy(&x, a + i, 8); memcpy(&y, b + i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

int pgorb_hamming_matrix(pgorb_ctx* c, const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out)
{
    if (!c) return PGORB_E_ARG;
    if (na < 0 || nb < 0 || (na && !a) || (nb && !b) || (na && nb && !out))
        return fail(c, PGORB_E_ARG, "bad argument to pgorb_hamming_matrix");
    if (!na || !nb) return 0;
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc;
    if ((rc = ensure(c, c->stageA, (size_t)na * 32))) return rc;
    if ((rc = ensure(c, c->stageB, (size_t)nb * 32))) return rc;
    if ((rc = ensure(c, c->stageOut, (size_t)na * nb * 2))) return rc;
    PG_HIP(c, hipMemcpy(c->stageA.p, a, (size_t)na * 32, hipMemcpyHostToDevice));
    PG_HIP(c, hipMemcpy(c->stageB.p, b, (size_t)nb * 32, hipMemcpyHostToDevice));
    pg_launch_hamming_matrix((uint8_t*)c->stageA.p, na, (uint8_t*)c->stageB.p, nb, (uint16_t*)c->stageOut.p, 0);
    PG_HIP(c, hipGetLastError());
    PG_HIP(c, hipMemcpy(out, c->stageOut.p, (size_t)na * nb * 2, hipMemcpyDeviceToHost));
    return 0;
}

int pgorb_hamming_best2(pgorb_ctx* c, const uint8_t* a, int na, const uint8_t* b, int nb,
                        int32_t* best_idx, uint16_t* best, uint16_t* second)
{
    if (!c) return PGORB_E_ARG;
    if (na < 0 || nb < 0 || (na && (!a || !best_idx || !best || !second)) || (nb && !b))
        return fail(c, PGORB_E_ARG, "bad argument to pgorb_hamming_best2");
    if (nb >= (1 << 20)) return fail(c, PGORB_E_LIMIT, "more than 2^20 train descriptors");
    if (!na) return 0;
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc;
    if ((rc = ensure(c, c->stageA, (size_t)na * 32))) return rc;
    if ((rc = ensure(c, c->stageB, (size_t)nb * 32 + 32))) return rc;
    if ((rc = ensure(c, c->stageOut, (size_t)na * 8 + 64))) return rc;
    PG_HIP(c, hipMemcpy(c->stageA.p, a, (size_t)na * 32, hipMemcpyHostToDevice));
    if (nb) PG_HIP(c, hipMemcpy(c->stageB.p, b, (size_t)nb * 32, hipMemcpyHostToDevice));
    int32_t* d_idx = (int32_t*)c->stageOut.p;
    uint16_t* d_b1 = (uint16_t*)(d_idx + na);
    uint16_t* d_b2 = d_b1 + na;
    if ((rc = ensure(c, c->xdesc, pg_match_scratch_bytes(c->mx, nb, 1) + 16))) return rc;
    pg_launch_best2(c->mx, (uint8_t*)c->stageA.p, na, (uint8_t*)c->stageB.p, nb, (uint8_t*)c->xdesc.p, d_idx, d_b1, d_b2, 0);
    PG_HIP(c, hipGetLastError());
    PG_HIP(c, hipMemcpy(best_idx, d_idx, (size_t)na * 4, hipMemcpyDeviceToHost));
    PG_HIP(c, hipMemcpy(best, d_b1, (size_t)na * 2, hipMemcpyDeviceToHost));
    PG_HIP(c, hipMemcpy(second, d_b2, (size_t)na * 2, hipMemcpyDeviceToHost));
    return 0;
}

int pgorb_match_batch_device(pgorb_ctx* c, const uint8_t* d_desc, const int32_t* d_n, int cap,
                             const int32_t* d_pq, const int32_t* d_pt, int npairs,
                             int32_t* d_best_idx, uint16_t* d_best, uint16_t* d_second, void* stream)
{
    if (!c) return PGORB_E_ARG;
    if (!d_desc || !d_n || cap < 1 || npairs < 0 || (npairs && (!d_pq || !d_pt || !d_best_idx || !d_best || !d_second)))
        return fail(c, PGORB_E_ARG, "bad argument to pgorb_match_batch_device");
    PG_HIP(c, hipSetDevice(c->prm.device));
    if (cap >= (1 << 20)) return fail(c, PGORB_E_LIMIT, "more than 2^20 descriptors per frame");
    int rc;
    if ((rc = ensure(c, c->xdesc, pg_match_scratch_bytes(c->mx, cap, npairs) + 16))) return rc;
    hipEvent_t* ev = (c->profMatch < c->profMax) ? &c->evMatch[2 * (size_t)c->profMatch] : nullptr;
    if (ev) PG_HIP(c, hipEventRecord(ev[0], (hipStream_t)stream));
    pg_launch_match_batch(c->mx, d_desc, d_n, cap, d_pq, d_pt, npairs, (uint8_t*)c->xdesc.p, d_best_idx, d_best, d_second,
                          (hipStream_t)stream);
    if (ev) { PG_HIP(c, hipEventRecord(ev[1], (hipStream_t)stream)); c->profMatch++; }
    PG_HIP(c, hipGetLastError());
    return 0;
}

int pgorb_profile_begin(pgorb_ctx* c, int max_calls)
{
    if (!c || max_calls < 0) return PGORB_E_ARG;
    PG_HIP(c, hipSetDevice(c->prm.device));
    while ((int)c->evExtract.size() < 5 * max_calls) {
        hipEvent_t e; PG_HIP(c, hipEventCreate(&e)); c->evExtract.push_back(e);
    }
    while ((int)c->evMatch.size() < 2 * max_calls) {
        hipEvent_t e; PG_HIP(c, hipEventCreate(&e)); c->evMatch.push_back(e);
    }
    c->profMax = max_calls; c->profExtract = 0; c->profMatch = 0;
    return 0;
}

int pgorb_set_option(pgorb_ctx* c, const char* key, int value)
{
    if (!key) return PGORB_E_ARG;
    if (!c) return PGORB_E_ARG;                               // every option belongs to a context (round 4: no process-wide state)
    pg_forward_option_to_lanes(c, key, value);                // the sibling contexts of live multi-lane device streams follow
    c->planEpoch++;                                            // (a captured graph of the host-frame path holds the old settings)
    if (!strcmp(key, "matcher")) { c->mx.popcount = value ? 1 : pg_match_default_opts().popcount; return 0; }     // 0 = what the environment says
    if (!strcmp(key, "match_mode")) {
        if (value < -1 || value > 2) return fail(c, PGORB_E_ARG, "match_mode must be -1, 0, 1 or 2");
        c->mx.mode = value;
        return 0;
    }
    if (!strcmp(key, "fast_tile_pitch")) {
        if (value != 0 && (value < 48 || value > 128 || value % 16)) return fail(c, PGORB_E_ARG, "fast_tile_pitch must be 0 (automatic) or 48, 64, ... 128");
        c->fastTilePitch = value; c->plan.fastTilePitch = value;
        return 0;
    }
    if (!strcmp(key, "fast_waves_per_block")) {
        if (value != 1 && value != 2 && value != 4) return fail(c, PGORB_E_ARG, "fast_waves_per_block must be 1, 2 or 4");
        c->fastWpb = value; c->plan.fastWpb = value;
        return 0;
    }
    if (!strcmp(key, "fast_cells_per_wave")) {
        if (value < 1 || value > 64) return fail(c, PGORB_E_ARG, "fast_cells_per_wave must be 1 ... 64");
        c->fastCpw = value; c->plan.fastCpw = value;
        return 0;
    }
    if (!strcmp(key, "quadtree_threads")) {
        if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(c, PGORB_E_ARG, "quadtree_threads must be 0 (automatic), 256, 512 or 1024");
        c->qtThreads = value; c->plan.qtThreads = value;
        return 0;
    }
    if (!strcmp(key, "quadtree_split")) {
        if (value < 0 || value > 2) return fail(c, PGORB_E_ARG, "quadtree_split must be 0 (one launch), 1 (two) or 2 (automatic)");
        c->qtSplit = value; c->plan.qtSplit = value;
        return 0;
    }
    if (!strcmp(key, "fused_levels")) { c->fused = value ? 1 : 0; c->fuse.enabled = c->fused; return 0; }
    if (!strcmp(key, "pipeline_pyramid")) { c->pipePyr = value ? 1 : 0; return 0; }
    if (!strcmp(key, "pipeline_levels")) { c->pipeLev = value & ((1 << PG_MAXL) - 2); return 0; }
    if (!strcmp(key, "pipeline_levels_priority")) { c->pipeLevPrio = value ? 1 : 0; return 0; }
    return fail(c, PGORB_E_ARG, "unknown option '%s'", key);
}

int pgorb_get_option(const pgorb_ctx* c, const char* key)
{
    if (!key || !c) return PGORB_OPTION_UNKNOWN;
    if (!strcmp(key, "fast_tile_pitch")) return c->fastTilePitch;
    if (!strcmp(key, "fast_waves_per_block")) return c->fastWpb;
    if (!strcmp(key, "fast_cells_per_wave")) return c->fastCpw;
    if (!strcmp(key, "quadtree_split")) return c->qtSplit;
    if (!strcmp(key, "quadtree_threads")) return c->qtThreads;
    if (!strcmp(key, "matcher")) return c->mx.popcount;
    if (!strcmp(key, "match_mode")) return c->mx.mode;
    if (!strcmp(key, "fused_levels")) return c->fused;
    if (!strcmp(key, "pipeline_pyramid")) return c->pipePyr;
    if (!strcmp(key, "pipeline_levels")) return c->pipeLev;
    if (!strcmp(key, "pipeline_levels_priority")) return c->pipeLevPrio;
    return PGORB_OPTION_UNKNOWN;
}

int pgorb_matcher_is_popcount(const pgorb_ctx* c, int cap_per_frame) { return pg_match_uses_popcount(c ? c->mx : pg_match_default_opts(), cap_per_frame) ? 1 : 0; }

int pgorb_profile_read(pgorb_ctx* c, double* ms)
{
    if (!c || !ms) return PGORB_E_ARG;
    PG_HIP(c, hipSetDevice(c->prm.device));
    PG_HIP(c, hipDeviceSynchronize());
    for (int i = 0; i < PGORB_NSTAGES; i++) ms[i] = 0;
    for (int k = 0; k < c->profExtract; k++)
        for (int st = 0; st < 4; st++) {
            float t = 0;
            PG_HIP(c, hipEventElapsedTime(&t, c->evExtract[5 * (size_t)k + st], c->evExtract[5 * (size_t)k + st + 1]));
            ms[st] += t;
        }
    for (int k = 0; k < c->profMatch; k++) {
        float t = 0;
        PG_HIP(c, hipEventElapsedTime(&t, c->evMatch[2 * (size_t)k], c->evMatch[2 * (size_t)k + 1]));
        ms[4] += t;
    }
    for (int st = 0; st < 4; st++) if (c->profExtract) ms[st] /= c->profExtract;
    if (c->profMatch) ms[4] /= c->profMatch;
    const int n = c->profExtract;
    c->profMax = 0;
    return n;
}

// ---- stage taps -----------------------------------------------------------------------------
int pgorb_debug_level_size(const pgorb_ctx* c, int level, int* w, int* h)
{
    if (!c || !c->planValid || level < 0 || level >= c->prm.nlevels) return PGORB_E_ARG;
    *w = c->plan.lvl[level].w; *h = c->plan.lvl[level].h;
    return 0;
}

int pgorb_debug_level_image(pgorb_ctx* c, int frame, int level, uint8_t* out)
{
    if (!c || !c->planValid || level < 0 || level >= c->prm.nlevels || frame < 0 || frame >= c->lastFrames)
        return PGORB_E_ARG;
    if (level == 0 && c->lastAliased) return fail(c, PGORB_E_ARG, "level 0 aliased the caller's buffer");
    const PgLevel& V = c->plan.lvl[level];
    PG_HIP(c, hipSetDevice(c->prm.device));
    PG_HIP(c, hipDeviceSynchronize());
    PG_HIP(c, hipMemcpy2D(out, V.w, V.img + (int64_t)frame * V.fstride, V.pitch, V.w, V.h, hipMemcpyDeviceToHost));
    return 0;
}

int pgorb_debug_level_candidates(pgorb_ctx* c, int frame, int level, int32_t* x, int32_t* y,
                                 int32_t* response, int cap)
{
    if (!c || !c->planValid || level < 0 || level >= c->prm.nlevels || frame < 0 || frame >= c->lastFrames)
        return PGORB_E_ARG;
    const PgPlan& P = c->plan;
    PG_HIP(c, hipSetDevice(c->prm.device));
    PG_HIP(c, hipDeviceSynchronize());
    // K2's per-cell slots of the level (K3 reads them in place since round 4: no dense candidate records exist any more)
    const PgLevel& V = P.lvl[level];
    const int ncells = V.nCols * V.nRows;
    std::vector<int32_t> cc(ncells);
    std::vector<uint32_t> slots((size_t)ncells * V.cellCap);
    PG_HIP(c, hipMemcpy(cc.data(), P.cellCount + (int64_t)frame * P.totalCells + V.cellBase, (size_t)ncells * 4, hipMemcpyDeviceToHost));
    PG_HIP(c, hipMemcpy(slots.data(), P.cellCand + (int64_t)frame * P.cellCandFrame + V.cellCandOff, slots.size() * 4, hipMemcpyDeviceToHost));
    int cnt = 0;
    for (int ci = 0; ci < ncells; ci++)
        for (int j = 0; j < std::min(cc[ci], V.cellCap); j++, cnt++) {
            if (cnt >= cap) continue;
            const uint32_t v = slots[(size_t)ci * V.cellCap + j];
            x[cnt] = v & 0xFFF; y[cnt] = (v >> 12) & 0xFFF; response[cnt] = v >> 24;
        }
    int32_t k3 = 0;                                            // K3's own count of the same slots must agree
    PG_HIP(c, hipMemcpy(&k3, P.candCount + frame * PG_MAXL + level, 4, hipMemcpyDeviceToHost));
    if (k3 != cnt) return fail(c, PGORB_E_OVERFLOW, "level %d: K3 counted %d candidates, the cell slots hold %d", level, k3, cnt);
    return cnt;
}

int pgorb_debug_level_keypoints(pgorb_ctx* c, int frame, int level)
{
    if (!c || !c->planValid || level < 0 || level >= c->prm.nlevels || frame < 0 || frame >= c->lastFrames)
        return PGORB_E_ARG;
    PG_HIP(c, hipSetDevice(c->prm.device));
    PG_HIP(c, hipDeviceSynchronize());
    int32_t cnt = 0;
    PG_HIP(c, hipMemcpy(&cnt, c->plan.kpCount + frame * PG_MAXL + level, 4, hipMemcpyDeviceToHost));
    return cnt;
}

}  // extern "C"

// ---- streamed ingest: frames that start in HOST memory ----------------------------------------------
// The reference's frames come from the decoder one at a time (src/io/image_sequence_reader.cc:138-208) and are
// consumed by the tracking loop (src/slam/track_image_sequence.cc:43-47).  Here the decoder writes grey frames
// straight into page-locked input slots; a slot (one batch) then flows through three HIP streams --
//   copy-in:  H2D of the slot's frames                        (PCIe, ~2.07 MB per 1080p frame)
//   compute:  K1..K6 on the slot's device frames (level 0 aliases them) + K7 of every frame against its
//             predecessor, including the last frame of the previous batch
//   copy-out: D2H of counts, keypoints, descriptors, matches into the slot's page-locked result block
// so that the upload of batch i+1 and the download of batch i-1 overlap the kernels of batch i.  Events order
// the three streams per slot; nothing blocks the host until pgorb_stream_wait.
struct pgorb_stream {
    pgorb_ctx* c = nullptr;
    int w = 0, h = 0, B = 0, depth = 0, cap = 0;               // w x h: the UPRIGHT frame the extractor sees
    // input format of the slots (pgorb_stream_create_ingest): frames exactly as decoded -- srcW x srcH pixels of `ch`
    // interleaved bytes, rotation / flips / grey conversion done on the device in front of K1 (k_ingest*, pyramid.hip)
    int srcW = 0, srcH = 0, ch = 1, rgbOrder = 1, rot = 0, vflip = 0, hflip = 0;
    bool ingest = false;                                      // false: grey, upright -> level 0 aliases the slot's device frames
    size_t inBytes = 0;                                       // bytes per input frame
    bool dead = false;                                        // a submit failed half way: the stream only accepts destroy
    hipStream_t sIn = nullptr, sRun = nullptr, sOut = nullptr;
    // Device-resident form (pgorb_stream_create_device, round 5): frames come from the caller's device memory, results stay
    // on the device, and consecutive batches run on `lanes` independent extractor working sets (lane 0 = the context itself,
    // the others private sibling contexts with the same parameters and options), each on its own HIP stream -- two batches
    // in flight let K1 (HBM), K2 / K4-6 (VALU issue), K3 (latency) and K7 (matrix pipe) of neighbouring batches share the chip.
    // Slot k runs on lane k % lanes.  What crosses batches -- the previous batch's last frame for the first match, the
    // front-end stage's state -- is one short section per batch; the sections run in submission order on a stream of their
    // own (sChain), each behind its batch's K1..K6, so the lanes never wait for each other: they drift apart and kernels of
    // DIFFERENT kinds end up side by side (a first build queued the section at the end of the lane's own stream, chained by
    // an event: the lanes then ran in lockstep, K1 beside K1 and K2 beside K2, and gained nothing -- 98.1 k against 98.3 k).
    bool device = false;
    std::vector<pgorb_ctx*> lane;          // [0] = c
    std::vector<hipStream_t> sLane;        // [0] = sRun
    int stagger = 0;                       // 0: lanes start their batches as soon as they can; 1: a batch's K1 starts when the previously
                                           // submitted batch has finished ITS K1 (its K2 is starting); 2: ... has finished its K2
    int lastLane = -1;                     // the lane the previously submitted batch runs on
    hipStream_t sChain = nullptr;          // the sections that cross batches, in submission order (several lanes: a stream of its own)
    int32_t* hStatus = nullptr;            // pinned, one word per slot (device form: the batch's status word)
    struct Slot {
        uint8_t* hIn = nullptr;            // pinned [B][srcH][srcW][ch]
        uint8_t* dIn = nullptr;            // device copy of it
        uint8_t* dOut = nullptr;           // device result block (layout below)
        uint8_t* hOut = nullptr;           // pinned copy of it
        hipEvent_t evIn = nullptr, evRun = nullptr, evOut = nullptr, evExt = nullptr;   // evExt: K1..K6 of the slot's batch done (device form)
        int frames = 0; bool busy = false;
    };
    std::vector<Slot> slot;
    // result block: n[B+1] (index 0 = the previous batch's last frame) | kps[B][cap] | desc[B+1][cap][32] |
    // best_idx[B][cap] | best[B][cap] | second[B][cap]
    size_t offN = 0, offK = 0, offD = 0, offI = 0, offB1 = 0, offB2 = 0, outBytes = 0;
    int32_t *dPq = nullptr, *dPt = nullptr;                // pairs (f, f-1), f = 1..B, in desc[] indexing
    uint8_t* dPrevDesc = nullptr; int32_t* dPrevN = nullptr; pgorb_keypoint* dPrevKps = nullptr;
    bool havePrev = false;
    // optional front-end stage (pgorb_stream_frontend): + matches12[B][cap] | nmatches[B] | word[B][cap] | weight[B][cap] | node[B][cap]
    bool fe = false; int feWindow = 100, feCheckOri = 1, feLevelsUp = -1; float feRatio = 0.9f, feBounds[4] = {0, 0, 0, 0};
    size_t offM12 = 0, offNM = 0, offW = 0, offWt = 0, offNd = 0;
    int32_t *dGridStart = nullptr, *dGridIdx = nullptr; float* dPrevMatched = nullptr;     // device scratch, [B+1] frames
};

static int stream_submit_queue(pgorb_stream* s, pgorb_stream::Slot& sl, int nframes, int slotIndex = 0, const uint8_t* d_frames = nullptr,
                               int stride = 0, int64_t frame_stride = 0, hipStream_t caller = nullptr);

// result-block layout for the stream's current settings (kps and desc hold B+1 frames: index 0 = the previous batch's last frame)
static void stream_layout(pgorb_stream* s)
{
    const size_t cap = (size_t)s->cap, B = (size_t)s->B;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    s->offN = 0; s->offK = al((B + 1) * 4); s->offD = s->offK + al((B + 1) * cap * sizeof(pgorb_keypoint));
    s->offI = s->offD + al((B + 1) * cap * 32); s->offB1 = s->offI + al(B * cap * 4);
    s->offB2 = s->offB1 + al(B * cap * 2); s->outBytes = s->offB2 + al(B * cap * 2);
    if (s->fe) {
        s->offM12 = s->outBytes; s->offNM = s->offM12 + al(B * cap * 4); s->outBytes = s->offNM + al(B * 4);
        if (s->feLevelsUp >= 0) {
            s->offWt = s->outBytes; s->offW = s->offWt + al(B * cap * 8); s->offNd = s->offW + al(B * cap * 4);
            s->outBytes = s->offNd + al(B * cap * 4);
        }
    }                                                        // (+ the status word behind it)
}

extern "C" {

int pgorb_stream_create_ingest(pgorb_ctx* c, int src_w, int src_h, int channels, int rgb_order, int rotate_degrees,
                               int vertical_flip, int horizontal_flip, int batch, int depth, pgorb_stream** out)
{
    if (!c || !out) return PGORB_E_ARG;
    *out = nullptr;
    if (batch < 1 || batch > c->prm.max_batch || depth < 2 || depth > 8 || src_w < 1 || src_h < 1)
        return fail(c, PGORB_E_ARG, "pgorb_stream_create: batch 1..max_batch, depth 2..8");
    if (channels != 1 && channels != 3 && channels != 4)
        return fail(c, PGORB_E_ARG, "pgorb_stream_create_ingest: channels must be 1, 3 or 4");
    if (rotate_degrees != 0 && rotate_degrees != 90 && rotate_degrees != 180 && rotate_degrees != 270)
        return fail(c, PGORB_E_ARG, "unsupported rotation %d: only multiples of 90 degrees", rotate_degrees);   // reader :203-207
    const bool swap = rotate_degrees == 90 || rotate_degrees == 270;
    const int w = swap ? src_h : src_w, h = swap ? src_w : src_h;
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc = make_plan(c, w, h, batch);
    if (rc) return rc;
    pgorb_stream* s = new pgorb_stream();
    s->c = c; s->w = w; s->h = h; s->B = batch; s->depth = depth; s->cap = c->plan.selTotal;
    s->srcW = src_w; s->srcH = src_h; s->ch = channels; s->rgbOrder = rgb_order ? 1 : 0; s->rot = rotate_degrees / 90;
    s->vflip = vertical_flip ? 1 : 0; s->hflip = horizontal_flip ? 1 : 0;
    s->ingest = channels != 1 || s->rot || s->vflip || s->hflip;
    s->inBytes = (size_t)src_w * src_h * channels;
    const size_t cap = (size_t)s->cap, B = (size_t)batch;
    stream_layout(s);
    bool ok = hipStreamCreateWithFlags(&s->sIn, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&s->sRun, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&s->sOut, hipStreamNonBlocking) == hipSuccess;
    s->slot.resize(depth);
    {
        // the page-locked input slots are the expensive part of a stream (0.3 ms per MB: 128 MB per slot of 64 1080p frames):
        // one thread per slot page-locks its buffers (round 4: the CLI's start-up; a third of the time with three slots)
        std::vector<int> okSlot(depth, 1);
        std::vector<std::thread> th;
        const int dev = c->prm.device;
        auto allocSlot = [&](int k) {
            pgorb_stream::Slot& sl = s->slot[k];
            bool o = hipSetDevice(dev) == hipSuccess;
            o = o && hipHostMalloc((void**)&sl.hIn, B * s->inBytes, hipHostMallocDefault) == hipSuccess;
            o = o && hipHostMalloc((void**)&sl.hOut, s->outBytes + 256, hipHostMallocDefault) == hipSuccess;
            okSlot[k] = o ? 1 : 0;
        };
        for (int k = 1; k < depth; k++) th.emplace_back(allocSlot, k);
        allocSlot(0);
        for (auto& t : th) t.join();
        for (int k = 0; k < depth; k++) ok = ok && okSlot[k];
    }
    for (auto& sl : s->slot) {
        ok = ok && hipMalloc((void**)&sl.dIn, B * s->inBytes + 256) == hipSuccess;
        ok = ok && hipMalloc((void**)&sl.dOut, s->outBytes + 256) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.evIn, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.evRun, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.evOut, hipEventDisableTiming) == hipSuccess;
    }
    std::vector<int32_t> pq(batch), pt(batch);
    for (int f = 0; f < batch; f++) { pq[f] = f + 1; pt[f] = f; }
    ok = ok && hipMalloc((void**)&s->dPq, B * 4) == hipSuccess && hipMalloc((void**)&s->dPt, B * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&s->dPrevDesc, cap * 32) == hipSuccess && hipMalloc((void**)&s->dPrevN, 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&s->dPrevKps, cap * sizeof(pgorb_keypoint)) == hipSuccess;
    ok = ok && hipMemcpy(s->dPq, pq.data(), B * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(s->dPt, pt.data(), B * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && ensure(c, c->xdesc, pg_match_scratch_bytes(c->mx, s->cap, batch) + 16) == 0;
    if (!ok) { pgorb_stream_destroy(s); return fail(c, PGORB_E_HIP, "pgorb_stream_create: allocation failed"); }
    s->lane.assign(1, c); s->sLane.assign(1, s->sRun); s->sChain = s->sRun;
    c->streams.push_back(s);
    *out = s;
    return 0;
}

// The device-resident form: `lanes` extractor working sets behind one stream object (include/pgorb.h).
void pg_forward_option_to_lanes(pgorb_ctx* c, const char* key, int value)
{
    for (pgorb_stream* st : c->streams)
        for (size_t l = 1; l < st->lane.size(); l++)
            if (st->lane[l] && st->lane[l] != c) (void)pgorb_set_option(st->lane[l], key, value);
}

// every tunable of a context (pgorb_set_option), for the sibling contexts of a multi-lane device stream: "same parameters and
// options" (pgorb.h) -- one list, used at lane creation; pgorb_set_option forwards later changes to the lanes of live streams
static void copy_tunables(pgorb_ctx* dst, const pgorb_ctx* src)
{
    dst->mx = src->mx; dst->qtThreads = src->qtThreads; dst->qtSplit = src->qtSplit; dst->fastTilePitch = src->fastTilePitch;
    dst->fastWpb = src->fastWpb; dst->fastCpw = src->fastCpw; dst->fused = src->fused; dst->pipePyr = src->pipePyr; dst->pipeLev = src->pipeLev;
    dst->pipeLevPrio = src->pipeLevPrio;
}

int pgorb_stream_create_device(pgorb_ctx* c, int w, int h, int batch, int depth, int lanes, pgorb_stream** out)
{
    if (!c || !out) return PGORB_E_ARG;
    *out = nullptr;
    if (batch < 1 || batch > c->prm.max_batch || depth < 2 || depth > 8 || lanes < 1 || lanes > depth || w < 1 || h < 1)
        return fail(c, PGORB_E_ARG, "pgorb_stream_create_device: batch 1..max_batch, depth 2..8, lanes 1..depth");
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc = make_plan(c, w, h, batch);
    if (rc) return rc;
    pgorb_stream* s = new pgorb_stream();
    s->c = c; s->w = w; s->h = h; s->B = batch; s->depth = depth; s->cap = c->plan.selTotal;
    s->srcW = w; s->srcH = h; s->ch = 1; s->device = true;
    s->inBytes = (size_t)w * h;
    const size_t cap = (size_t)s->cap, B = (size_t)batch;
    stream_layout(s);
    s->lane.assign(1, c);
    bool ok = hipStreamCreateWithFlags(&s->sRun, hipStreamNonBlocking) == hipSuccess;
    s->sLane.assign(1, s->sRun);
    for (int l = 1; l < lanes && ok; l++) {
        // a sibling context: the same extractor (parameters, options), its own pyramid / candidate / selection arenas and plan
        pgorb_ctx* lc = nullptr;
        hipStream_t ls = nullptr;
        ok = pgorb_create(&c->prm, &lc) == PGORB_OK;
        if (ok) {
            copy_tunables(lc, c);
            s->lane.push_back(lc);
            ok = make_plan(lc, w, h, batch) == 0 && hipStreamCreateWithFlags(&ls, hipStreamNonBlocking) == hipSuccess;
            s->sLane.push_back(ls);
        }
    }
    s->slot.resize(depth);
    ok = ok && hipHostMalloc((void**)&s->hStatus, 64 * sizeof(int32_t), hipHostMallocDefault) == hipSuccess;
    if (lanes > 1) ok = ok && hipStreamCreateWithFlags(&s->sChain, hipStreamNonBlocking) == hipSuccess;
    else s->sChain = s->sRun;
    if (const char* e = getenv("PGORB_STREAM_STAGGER")) s->stagger = atoi(e);
    if (lanes > 1)
        for (pgorb_ctx* lc : s->lane) {
            if (!lc->evPyrEnd) ok = ok && hipEventCreateWithFlags(&lc->evPyrEnd, hipEventDisableTiming) == hipSuccess;
            if (!lc->evFastEnd) ok = ok && hipEventCreateWithFlags(&lc->evFastEnd, hipEventDisableTiming) == hipSuccess;
        }
    for (auto& sl : s->slot) {
        ok = ok && hipMalloc((void**)&sl.dOut, s->outBytes + 256) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.evIn, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.evRun, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.evExt, hipEventDisableTiming) == hipSuccess;
    }
    std::vector<int32_t> pq(batch), pt(batch);
    for (int f = 0; f < batch; f++) { pq[f] = f + 1; pt[f] = f; }
    ok = ok && hipMalloc((void**)&s->dPq, B * 4) == hipSuccess && hipMalloc((void**)&s->dPt, B * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&s->dPrevDesc, cap * 32) == hipSuccess && hipMalloc((void**)&s->dPrevN, 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&s->dPrevKps, cap * sizeof(pgorb_keypoint)) == hipSuccess;
    ok = ok && hipMemcpy(s->dPq, pq.data(), B * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(s->dPt, pt.data(), B * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && ensure(c, c->xdesc, pg_match_scratch_bytes(c->mx, s->cap, batch) + 16) == 0;
    if (!ok) { pgorb_stream_destroy(s); return fail(c, PGORB_E_HIP, "pgorb_stream_create_device: allocation failed"); }
    c->streams.push_back(s);
    *out = s;
    return 0;
}

int pgorb_stream_create(pgorb_ctx* c, int w, int h, int batch, int depth, pgorb_stream** out)
{
    return pgorb_stream_create_ingest(c, w, h, 1, 1, 0, 0, 0, batch, depth, out);
}

void pgorb_stream_destroy(pgorb_stream* s)
{
    if (!s) return;
    (void)hipSetDevice(s->c->prm.device);
    {
        auto& v = s->c->streams;
        v.erase(std::remove(v.begin(), v.end(), s), v.end());
    }
    if (s->sIn) (void)hipStreamSynchronize(s->sIn);
    if (s->sRun) (void)hipStreamSynchronize(s->sRun);
    if (s->sOut) (void)hipStreamSynchronize(s->sOut);
    for (size_t l = 1; l < s->sLane.size(); l++) if (s->sLane[l]) { (void)hipStreamSynchronize(s->sLane[l]); (void)hipStreamDestroy(s->sLane[l]); }
    for (size_t l = 1; l < s->lane.size(); l++) pgorb_destroy(s->lane[l]);      // the private sibling contexts
    if (s->sChain && s->lane.size() > 1) { (void)hipStreamSynchronize(s->sChain); (void)hipStreamDestroy(s->sChain); }
    if (s->hStatus) (void)hipHostFree(s->hStatus);
    for (auto& sl : s->slot) {
        if (sl.hIn) (void)hipHostFree(sl.hIn);
        if (sl.dIn) (void)hipFree(sl.dIn);
        if (sl.dOut) (void)hipFree(sl.dOut);
        if (sl.hOut) (void)hipHostFree(sl.hOut);
        if (sl.evIn) (void)hipEventDestroy(sl.evIn);
        if (sl.evRun) (void)hipEventDestroy(sl.evRun);
        if (sl.evOut) (void)hipEventDestroy(sl.evOut);
        if (sl.evExt) (void)hipEventDestroy(sl.evExt);
    }
    if (s->dPq) (void)hipFree(s->dPq);
    if (s->dPt) (void)hipFree(s->dPt);
    if (s->dPrevDesc) (void)hipFree(s->dPrevDesc);
    if (s->dPrevN) (void)hipFree(s->dPrevN);
    if (s->dPrevKps) (void)hipFree(s->dPrevKps);
    if (s->dGridStart) (void)hipFree(s->dGridStart);
    if (s->dGridIdx) (void)hipFree(s->dGridIdx);
    if (s->dPrevMatched) (void)hipFree(s->dPrevMatched);
    if (s->sIn) (void)hipStreamDestroy(s->sIn);
    if (s->sRun) (void)hipStreamDestroy(s->sRun);
    if (s->sOut) (void)hipStreamDestroy(s->sOut);
    delete s;
}

uint8_t* pgorb_stream_input(pgorb_stream* s, int slot)
{
    return (s && slot >= 0 && slot < s->depth) ? s->slot[slot].hIn : nullptr;
}

int pgorb_stream_reset(pgorb_stream* s)                  // a new ride: the next batch has no predecessor frame
{
    if (!s) return PGORB_E_ARG;
    s->havePrev = false;
    return 0;
}

int pgorb_stream_lanes(const pgorb_stream* s) { return s ? (int)s->lane.size() : PGORB_E_ARG; }

int pgorb_stream_submit(pgorb_stream* s, int slot, int nframes)
{
    if (!s || slot < 0 || slot >= s->depth) return PGORB_E_ARG;
    pgorb_ctx* c = s->c;
    if (nframes < 1 || nframes > s->B) return fail(c, PGORB_E_ARG, "pgorb_stream_submit: 1..batch frames");
    if (s->dead) return fail(c, PGORB_E_HIP, "pgorb_stream_submit: an earlier submit failed half way; destroy the stream");
    if (s->device) return fail(c, PGORB_E_ARG, "pgorb_stream_submit: a device-resident stream takes pgorb_stream_submit_device");
    pgorb_stream::Slot& sl = s->slot[slot];
    if (sl.busy) return fail(c, PGORB_E_ARG, "pgorb_stream_submit: slot %d not collected with pgorb_stream_wait", slot);
    PG_HIP(c, hipSetDevice(c->prm.device));
    int rc = make_plan(c, s->w, s->h, nframes);
    if (rc) return rc;
    rc = stream_submit_queue(s, sl, nframes);
    if (rc) {
        // part of the batch may be queued on the three streams with no event recorded for the slot: drain them, so
        // that nothing is still writing into the slot's buffers, and retire the stream
        (void)hipStreamSynchronize(s->sIn); (void)hipStreamSynchronize(s->sRun); (void)hipStreamSynchronize(s->sOut);
        s->dead = true;
        return rc;
    }
    sl.frames = nframes; sl.busy = true;
    return 0;
}

int pgorb_stream_submit_device(pgorb_stream* s, int slot, const uint8_t* d_frames, int nframes, int stride, int64_t frame_stride,
                               void* hip_stream)
{
    if (!s || slot < 0 || slot >= s->depth) return PGORB_E_ARG;
    pgorb_ctx* c = s->c;
    if (!s->device) return fail(c, PGORB_E_ARG, "pgorb_stream_submit_device: the stream was created for host frames");
    if (nframes < 1 || nframes > s->B || !d_frames || stride < s->w) return fail(c, PGORB_E_ARG, "pgorb_stream_submit_device: 1..batch frames, stride >= width");
    if (s->dead) return fail(c, PGORB_E_HIP, "pgorb_stream_submit_device: an earlier submit failed half way; destroy the stream");
    pgorb_stream::Slot& sl = s->slot[slot];
    if (sl.busy) return fail(c, PGORB_E_ARG, "pgorb_stream_submit_device: slot %d not collected with pgorb_stream_wait_device", slot);
    PG_HIP(c, hipSetDevice(c->prm.device));
    pgorb_ctx* lc = s->lane[slot % (int)s->lane.size()];
    int rc = make_plan(lc, s->w, s->h, nframes);
    if (rc) { if (lc != c) c->err = lc->err; return rc; }
    rc = stream_submit_queue(s, sl, nframes, slot, d_frames, stride, frame_stride, (hipStream_t)hip_stream);
    if (rc) {
        for (hipStream_t q : s->sLane) (void)hipStreamSynchronize(q);
        (void)hipStreamSynchronize(s->sChain);
        s->dead = true;
        return rc;
    }
    sl.frames = nframes; sl.busy = true;
    return 0;
}

}  // extern "C"

static int stream_submit_queue(pgorb_stream* s, pgorb_stream::Slot& sl, int nframes, int slotIndex, const uint8_t* d_frames,
                               int stride, int64_t frame_stride, hipStream_t caller)
{
    pgorb_ctx* c = s->c;                                       // errors, the vocabulary and the matchers' scratch live here
    const int laneIx = s->device ? slotIndex % (int)s->lane.size() : 0;
    pgorb_ctx* lc = s->lane[laneIx];                           // the extractor working set this batch runs on
    hipStream_t sr = s->sLane[laneIx];
    int rc;
    const size_t fbytes = s->inBytes, cap = (size_t)s->cap;
    if (!s->device) {
        // copy-in: after the kernels of this slot's previous batch have read its device frames
        PG_HIP(c, hipStreamWaitEvent(s->sIn, sl.evRun, 0));
        PG_HIP(c, hipMemcpyAsync(sl.dIn, sl.hIn, fbytes * nframes, hipMemcpyHostToDevice, s->sIn));
        PG_HIP(c, hipEventRecord(sl.evIn, s->sIn));
        // compute: after the upload, and after the previous download of this slot's result block
        PG_HIP(c, hipStreamWaitEvent(sr, sl.evIn, 0));
        PG_HIP(c, hipStreamWaitEvent(sr, sl.evOut, 0));
    } else {
        // the caller's frames are ready where the caller's stream stands now; the slot's result block is free once the
        // section of the batch that used it last is done (a never-recorded event does not wait)
        PG_HIP(c, hipEventRecord(sl.evIn, caller));
        PG_HIP(c, hipStreamWaitEvent(sr, sl.evIn, 0));
        PG_HIP(c, hipStreamWaitEvent(sr, sl.evRun, 0));
        // stagger (PGORB_STREAM_STAGGER=1|2 at stream creation; off by default): hold a batch back until the previously submitted
        // one (another lane) is past its pyramid (or past K2).  Measured on an MI355X (tools/experiments/r5_lanes.py,
        // profiles/r05_lanes.txt; 1080p / 2000, batch 128): one lane 94-96 k frames/s, two lanes 101.3 k without stagger, 101.1 k /
        // 99.0 k with 1 / 2, three lanes 99.0 k -- the hardware interleaves the lanes' kernels at workgroup granularity whatever
        // the start offsets, and K2's one-wave workgroups take every wave slot they can get.  A never-recorded event does not wait.
        if (s->stagger && s->lastLane >= 0 && s->lastLane != laneIx) {
            pgorb_ctx* pl = s->lane[s->lastLane];
            PG_HIP(c, hipStreamWaitEvent(sr, s->stagger == 2 ? pl->evFastEnd : pl->evPyrEnd, 0));
        }
        s->lastLane = laneIx;
    }
    int32_t* dN = (int32_t*)(sl.dOut + s->offN);
    pgorb_keypoint* dK = (pgorb_keypoint*)(sl.dOut + s->offK);
    uint8_t* dD = sl.dOut + s->offD;
    // ---- extraction: K1..K6 of this batch alone (nothing here looks at another batch) ----
    if (s->device) {
        rc = run_batch(lc, d_frames, false, nframes, s->w, s->h, stride, frame_stride, dK + cap, dD + cap * 32, s->cap, dN + 1, sr);
        if (rc && lc != c) c->err = lc->err;
    } else if (s->ingest) {
        // frames as decoded: rotation / flips / cvtColor on the device into level 0 (image_sequence_reader.cc:53-58,186-205;
        // Tracking.cc:247-260), then the extractor on the upright grey planes
        pg_launch_ingest(c->plan, sl.dIn, s->srcW * s->ch, (int64_t)fbytes, s->srcW, s->srcH, s->ch, s->rgbOrder, s->rot,
                         s->vflip != 0, s->hflip != 0, nframes, sr);
        rc = run_batch(c, nullptr, true, nframes, s->w, s->h, s->w, 0, dK + cap, dD + cap * 32, s->cap, dN + 1, sr);
    } else {
        rc = run_batch(c, sl.dIn, false, nframes, s->w, s->h, s->w, (int64_t)fbytes, dK + cap, dD + cap * 32, s->cap, dN + 1, sr);
    }
    if (rc) return rc;
    // the batch's device status word travels inside the result block (the lane's next batch resets the word)
    PG_HIP(c, hipMemcpyAsync(sl.dOut + s->outBytes, lc->plan.status, 4, hipMemcpyDeviceToDevice, sr));
    if (s->device) PG_HIP(c, hipMemcpyAsync(s->hStatus + slotIndex, lc->plan.status, 4, hipMemcpyDeviceToHost, sr));
    // ---- the section that crosses batches: in submission order on sChain, behind this batch's K1..K6 ----
    if (s->sChain != sr) {
        PG_HIP(c, hipEventRecord(sl.evExt, sr));
        PG_HIP(c, hipStreamWaitEvent(s->sChain, sl.evExt, 0));
        sr = s->sChain;
    }
    if (s->havePrev) {
        PG_HIP(c, hipMemcpyAsync(dD, s->dPrevDesc, cap * 32, hipMemcpyDeviceToDevice, sr));
        PG_HIP(c, hipMemcpyAsync(dN, s->dPrevN, 4, hipMemcpyDeviceToDevice, sr));
        if (s->fe) PG_HIP(c, hipMemcpyAsync(dK, s->dPrevKps, cap * sizeof(pgorb_keypoint), hipMemcpyDeviceToDevice, sr));
    } else {
        PG_HIP(c, hipMemsetAsync(dN, 0, 4, sr));
    }
    // the slab form (match_mode 0) may have been selected after the stream was created: size its arena for THIS launch
    // (ensure() only ever grows; hipFree of the old arena waits for the work that still uses it)
    if ((rc = ensure(c, c->xdesc, pg_match_scratch_bytes(c->mx, s->cap, nframes) + 16))) return rc;
    pg_launch_match_batch(c->mx, dD, dN, s->cap, s->dPq, s->dPt, nframes, (uint8_t*)c->xdesc.p, (int32_t*)(sl.dOut + s->offI),
                          (uint16_t*)(sl.dOut + s->offB1), (uint16_t*)(sl.dOut + s->offB2), sr);
    if (s->fe) {
        // what the tracking thread does with a fresh Frame, for the whole batch: the 64x48 grid of every frame
        // (Frame.cc:234-249), SearchForInitialization(previous, current) with vbPrevMatched = the previous frame's
        // keypoints (Tracking.cc:583-597), ORBVocabulary::transform of every descriptor (Frame.cc:399-406)
        const float* b = s->feBounds;
        if ((rc = pgorb_frame_grid_batch_device(c, dK + cap, dN + 1, nframes, s->cap, b[0], b[1], b[2], b[3],
                                                s->dGridStart + (PGORB_GRID_CELLS + 1), s->dGridIdx + cap, sr))) return rc;
        pg_launch_prev_matched_init(dK, (int64_t)nframes * cap, s->dPrevMatched, sr);
        if ((rc = pgorb_search_for_initialization_batch_device(c, dK, dD, dN, s->cap, s->dGridStart, s->dGridIdx, s->dPt, s->dPq, nframes,
                                                               b[0], b[1], b[2], b[3], s->dPrevMatched, (int32_t*)(sl.dOut + s->offM12),
                                                               (int32_t*)(sl.dOut + s->offNM), s->feWindow, s->feRatio, s->feCheckOri, sr))) return rc;
        if (s->feLevelsUp >= 0 &&
            (rc = pgorb_bow_transform_device(c, dD + cap * 32, nframes * s->cap, s->feLevelsUp, (uint32_t*)(sl.dOut + s->offW),
                                             (double*)(sl.dOut + s->offWt), (uint32_t*)(sl.dOut + s->offNd), sr))) return rc;
        PG_HIP(c, hipMemcpyAsync(s->dPrevKps, dK + (size_t)nframes * cap, cap * sizeof(pgorb_keypoint), hipMemcpyDeviceToDevice, sr));
    }
    PG_HIP(c, hipMemcpyAsync(s->dPrevDesc, dD + (size_t)nframes * cap * 32, cap * 32, hipMemcpyDeviceToDevice, sr));
    PG_HIP(c, hipMemcpyAsync(s->dPrevN, dN + nframes, 4, hipMemcpyDeviceToDevice, sr));
    PG_HIP(c, hipEventRecord(sl.evRun, sr));
    s->havePrev = true;
    if (!s->device) {
        // copy-out
        PG_HIP(c, hipStreamWaitEvent(s->sOut, sl.evRun, 0));
        PG_HIP(c, hipMemcpyAsync(sl.hOut, sl.dOut, s->outBytes + 4, hipMemcpyDeviceToHost, s->sOut));
        PG_HIP(c, hipEventRecord(sl.evOut, s->sOut));
    }
    PG_HIP(c, hipGetLastError());
    return 0;
}

extern "C" {

int pgorb_stream_wait(pgorb_stream* s, int slot, const int32_t** n, const pgorb_keypoint** kps, const uint8_t** desc,
                      const int32_t** best_idx, const uint16_t** best, const uint16_t** second, int* cap)
{
    if (!s || slot < 0 || slot >= s->depth) return PGORB_E_ARG;
    pgorb_ctx* c = s->c;
    pgorb_stream::Slot& sl = s->slot[slot];
    if (s->device) return fail(c, PGORB_E_ARG, "pgorb_stream_wait: a device-resident stream takes pgorb_stream_wait_device");
    if (!sl.busy) return fail(c, PGORB_E_ARG, "pgorb_stream_wait: slot %d has no batch in flight", slot);
    PG_HIP(c, hipSetDevice(c->prm.device));
    PG_HIP(c, hipEventSynchronize(sl.evOut));
    sl.busy = false;
    const int32_t st = *(const int32_t*)(sl.hOut + s->outBytes);      // the batch's device status word
    if (st) return fail(c, st, "device reported status %d", st);
    if (n) *n = (const int32_t*)(sl.hOut + s->offN) + 1;
    if (kps) *kps = (const pgorb_keypoint*)(sl.hOut + s->offK) + s->cap;
    if (desc) *desc = sl.hOut + s->offD + (size_t)s->cap * 32;
    if (best_idx) *best_idx = (const int32_t*)(sl.hOut + s->offI);
    if (best) *best = (const uint16_t*)(sl.hOut + s->offB1);
    if (second) *second = (const uint16_t*)(sl.hOut + s->offB2);
    if (cap) *cap = s->cap;
    return sl.frames;
}

// Device-resident form: the slot's batch is complete (host blocks on the batch's event, or -- hip_stream != NULL with
// wait_on_host == 0 -- that stream is made to wait for it and the call returns at once); DEVICE pointers into the slot's
// result block, valid until the slot is submitted again.
int pgorb_stream_wait_device(pgorb_stream* s, int slot, int wait_on_host, void* hip_stream, const int32_t** d_n, const pgorb_keypoint** d_kps,
                             const uint8_t** d_desc, const int32_t** d_best_idx, const uint16_t** d_best, const uint16_t** d_second, int* cap)
{
    if (!s || slot < 0 || slot >= s->depth) return PGORB_E_ARG;
    pgorb_ctx* c = s->c;
    if (!s->device) return fail(c, PGORB_E_ARG, "pgorb_stream_wait_device: the stream was created for host frames");
    pgorb_stream::Slot& sl = s->slot[slot];
    if (!sl.busy) return fail(c, PGORB_E_ARG, "pgorb_stream_wait_device: slot %d has no batch in flight", slot);
    PG_HIP(c, hipSetDevice(c->prm.device));
    if (wait_on_host) {                                       // (hip_stream == NULL is the legacy null stream, as in pgorb_stream_submit_device)
        PG_HIP(c, hipEventSynchronize(sl.evRun));
        const int32_t st = s->hStatus[slot];
        if (st) { sl.busy = false; return fail(c, st, "device reported status %d", st); }
    } else {
        PG_HIP(c, hipStreamWaitEvent((hipStream_t)hip_stream, sl.evRun, 0));        // (the status word: pgorb_check_async of the caller's choice)
    }
    sl.busy = false;
    if (d_n) *d_n = (const int32_t*)(sl.dOut + s->offN) + 1;
    if (d_kps) *d_kps = (const pgorb_keypoint*)(sl.dOut + s->offK) + s->cap;
    if (d_desc) *d_desc = sl.dOut + s->offD + (size_t)s->cap * 32;
    if (d_best_idx) *d_best_idx = (const int32_t*)(sl.dOut + s->offI);
    if (d_best) *d_best = (const uint16_t*)(sl.dOut + s->offB1);
    if (d_second) *d_second = (const uint16_t*)(sl.dOut + s->offB2);
    if (cap) *cap = s->cap;
    return sl.frames;
}

int pgorb_stream_frontend(pgorb_stream* s, float min_x, float max_x, float min_y, float max_y, int window_size, float nnratio,
                          int check_orientation, int bow_levelsup)
{
    if (!s) return PGORB_E_ARG;
    pgorb_ctx* c = s->c;
    if (!(max_x > min_x) || !(max_y > min_y) || window_size < 0) return fail(c, PGORB_E_ARG, "pgorb_stream_frontend: bounds / window");
    for (auto& sl : s->slot) if (sl.busy) return fail(c, PGORB_E_ARG, "pgorb_stream_frontend: a batch is in flight");
    if (s->cap > 16000) return fail(c, PGORB_E_LIMIT, "more than 16000 keypoints per frame");
    if (bow_levelsup >= 0) {
        const uint8_t* blob; int k, L, nn;
        int rc = pg_ctx_vocab_get(c, &blob, &k, &L, &nn);     // "no vocabulary resident" is reported here, not at the first submit
        if (rc) return rc;
    }
    PG_HIP(c, hipSetDevice(c->prm.device));
    PG_HIP(c, hipDeviceSynchronize());
    // new result blocks first; the stream's settings and buffers change only when every allocation succeeded
    pgorb_stream t = *s;                                       // (layout arithmetic on a copy)
    t.fe = true; t.feLevelsUp = bow_levelsup;
    stream_layout(&t);
    const size_t cap = (size_t)s->cap, B = (size_t)s->B;
    std::vector<uint8_t*> nd(s->slot.size(), nullptr), nh(s->slot.size(), nullptr);
    int32_t *gs = s->dGridStart, *gi = s->dGridIdx; float* pm = s->dPrevMatched;
    bool ok = true;
    for (size_t i = 0; i < s->slot.size(); i++) {
        ok = ok && hipMalloc((void**)&nd[i], t.outBytes + 256) == hipSuccess;
        ok = ok && (s->device || hipHostMalloc((void**)&nh[i], t.outBytes + 256, hipHostMallocDefault) == hipSuccess);
    }
    if (!s->dGridStart) {
        gs = nullptr; gi = nullptr; pm = nullptr;
        ok = ok && hipMalloc((void**)&gs, (B + 1) * (PGORB_GRID_CELLS + 1) * 4) == hipSuccess;
        ok = ok && hipMalloc((void**)&gi, (B + 1) * cap * 4) == hipSuccess;
        ok = ok && hipMalloc((void**)&pm, B * cap * 8) == hipSuccess;
    }
    if (!ok) {
        for (uint8_t* q : nd) if (q) (void)hipFree(q);
        for (uint8_t* q : nh) if (q) (void)hipHostFree(q);
        if (!s->dGridStart) { if (gs) (void)hipFree(gs); if (gi) (void)hipFree(gi); if (pm) (void)hipFree(pm); }
        return fail(c, PGORB_E_HIP, "pgorb_stream_frontend: allocation failed (the stream is unchanged)");
    }
    for (size_t i = 0; i < s->slot.size(); i++) {
        (void)hipFree(s->slot[i].dOut);
        if (s->slot[i].hOut) (void)hipHostFree(s->slot[i].hOut);
        s->slot[i].dOut = nd[i]; s->slot[i].hOut = nh[i];
    }
    s->dGridStart = gs; s->dGridIdx = gi; s->dPrevMatched = pm;
    s->fe = true; s->feWindow = window_size; s->feRatio = nnratio; s->feCheckOri = check_orientation ? 1 : 0; s->feLevelsUp = bow_levelsup;
    s->feBounds[0] = min_x; s->feBounds[1] = max_x; s->feBounds[2] = min_y; s->feBounds[3] = max_y;
    stream_layout(s);
    s->havePrev = false;
    return 0;
}

int pgorb_stream_frontend_results(pgorb_stream* s, int slot, const int32_t** matches12, const int32_t** nmatches,
                                  const uint32_t** word, const double** weight, const uint32_t** node)
{
    if (!s || slot < 0 || slot >= s->depth) return PGORB_E_ARG;
    pgorb_ctx* c = s->c;
    pgorb_stream::Slot& sl = s->slot[slot];
    if (!s->fe) return fail(c, PGORB_E_ARG, "pgorb_stream_frontend_results: the front-end stage is not enabled");
    if (sl.busy) return fail(c, PGORB_E_ARG, "pgorb_stream_frontend_results: collect slot %d with pgorb_stream_wait first", slot);
    const uint8_t* base = s->device ? sl.dOut : sl.hOut;      // (a device-resident stream hands out DEVICE pointers here as well)
    if (matches12) *matches12 = (const int32_t*)(base + s->offM12);
    if (nmatches) *nmatches = (const int32_t*)(base + s->offNM);
    const bool bow = s->feLevelsUp >= 0;
    if (word) *word = bow ? (const uint32_t*)(base + s->offW) : nullptr;
    if (weight) *weight = bow ? (const double*)(base + s->offWt) : nullptr;
    if (node) *node = bow ? (const uint32_t*)(base + s->offNd) : nullptr;
    return 0;
}

}  // extern "C"

// calib.hip -- K9: fit_motion's sliding-window accelerometer calibration, every window's L-BFGS
// fit running at once on the GPU (BASELINE.json configs[4]).
//
// Reference: for every window of 40 GPS fixes (step 5) fit_motion builds an AccelerometerCalibrator
// and minimises its 9-parameter loss with LBFGSpp (src/fit_motion.cc:151-246,
// src/calibration/velocity.cc:42-180, thirdparty/LBFGS/LBFGS.h:78-181, LBFGS/LineSearch.h:41-109);
// one loss evaluation integrates every IMU sample of the window in time order
// (src/geometry/geometry.cc:24-53), ~10^4 dependent steps, and the fits run one after another.
//
// Here: windows are independent, so ONE WORKGROUP PER WINDOW runs the whole solver, all windows at once.
// Every double is produced by the same operation on the same operands as in the reference (the sums
// stay sequential in time: reordering them changes the doubles and the JSON is diffed bit for
// bit); inside a window five waves form a pipeline over chunks of 128 steps (see the device section).
// What does not depend on the nine parameters is hoisted out of the solver and computed once per
// window on the host, with the same operations in the same order as the reference's loop body:
//   forward stream  (8 doubles / step): dt, orientation BEFORE the step (w,x,y,z), raw acceleration
//   backward stream (11 doubles / step): total_time_sec*dt, dt, dt * total_time_weighted_rotation^T (3x3)
//   per GPS interval: reference_distance, first step
// (RotationMotionToQuaternion's sin/cos therefore run in the host libm, as in the reference.)
// Streams are structure-of-arrays per window, so a chunk of 64 steps of one field is one coalesced
// 512-byte request.  fp64 add/mul/div/sqrt are IEEE on both sides, contraction is off (Makefile),
// so a workgroup reproduces the CPU run of its window bit for bit.
//
// Eigen reduction orders (the parity contract shared with oracle/calib_oracle.c, E0-E5 there):
// 3-vectors t0 + (t1 + t2); the solver's 9-vectors in SSE2 packet order.
#include "pgorb_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
#include <vector>

int pg_ctx_fail(pgorb_ctx* c, int code, const char* msg);
int pg_ctx_device(pgorb_ctx* c);

namespace {

#define CB_FWD 8
#define CB_BWD 11
#define CB_M 6            // LBFGSParam::m (Param.h:164)

struct Quat { double w, x, y, z; };

// ---- host: the reference's data preparation ----

inline Quat rotation_motion_to_quaternion(double rx, double ry, double rz, double duration_sec)   // geometry.cc:6-22
{
    const double rate = sqrt(rx * rx + ry * ry + rz * rz);
    const double half_theta = rate * duration_sec * 0.5;
    const double s = sin(half_theta) / (rate + 1e-30);
    return {cos(half_theta), rx * s, ry * s, rz * s};
}

__host__ __device__ inline double dot3(const double* a, const double* b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

__host__ __device__ inline void quat_rotate(const Quat& q, const double* v, double* out)     // Eigen _transformVector
{
    double uv0 = q.y * v[2] - q.z * v[1], uv1 = q.z * v[0] - q.x * v[2], uv2 = q.x * v[1] - q.y * v[0];
    uv0 += uv0; uv1 += uv1; uv2 += uv2;
    const double c0 = q.y * uv2 - q.z * uv1, c1 = q.z * uv0 - q.x * uv2, c2 = q.x * uv1 - q.y * uv0;
    out[0] = (v[0] + q.w * uv0) + c0; out[1] = (v[1] + q.w * uv1) + c1; out[2] = (v[2] + q.w * uv2) + c2;
}

inline Quat quat_mul(const Quat& a, const Quat& b)
{
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}

inline void quat_matrix(const Quat& q, double* R)                                              // Eigen toRotationMatrix
{
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z, twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

struct Imu {                                    // MergedTimeSeries over {rotation times, acceleration times}
    const double *rot, *acc;
    std::vector<int32_t> evRot, evAcc;          // MergedEvents()
    std::vector<int64_t> time;                  // MergedEventTimeUsec()
};

// align_time_series.cc:29-118.  false where the reference CHECK-fails or cannot merge.
bool merge_imu(const double* rot, const int64_t* rt, int nr, const double* acc, const int64_t* at, int na, Imu& M)
{
    M.rot = rot; M.acc = acc;
    if (nr <= 0 || na <= 0) return false;
    for (int i = 0; i + 1 < nr; i++) if (!(rt[i] < rt[i + 1])) return false;
    for (int i = 0; i + 1 < na; i++) if (!(at[i] < at[i + 1])) return false;
    const int64_t start = std::max(rt[0], at[0]), end = std::min(rt[nr - 1], at[na - 1]);
    if (end < start) return false;
    auto first = [&](const int64_t* t, int n) {
        const int idx = (int)(std::lower_bound(t, t + n, start) - t);
        return t[idx] > start ? idx - 1 : idx;
    };
    int ir = first(rt, nr), ia = first(at, na);
    if (ir < 0 || ia < 0) return false;
    for (;;) {
        M.evRot.push_back(ir); M.evAcc.push_back(ia); M.time.push_back(std::max(rt[ir], at[ia]));
        if (ir + 1 >= nr || ia + 1 >= na) break;
        const int64_t next = std::min(rt[ir + 1], at[ia + 1]);
        if (rt[ir + 1] == next) ir++;
        if (at[ia + 1] == next) ia++;
    }
    for (size_t i = 0; i + 1 < M.time.size(); i++) if (!(M.time[i] < M.time[i + 1])) return false;
    return true;
}

struct Step { int32_t event; int64_t usec; };   // one InterpolationInterval: its merged event and its duration

// One window = one AccelerometerCalibrator: reference_intervals_ flattened (align_time_series.cc:150-195).
struct Window {
    std::vector<int32_t> refCnt;                // steps per reference interval (refCnt[0] is always 0)
    std::vector<int32_t> refIdx;                // reference_end_time_index of each step's interval
    std::vector<Step> steps;
};

void make_window(const int64_t* ref_t, int n_ref, const Imu& M, Window& W)
{
    const std::vector<int64_t>& imu = M.time;
    const int n = (int)imu.size();
    W.refCnt.assign(n_ref, 0);
    // reference_idx == 0 pushes nothing: it only advances the cursor past every sample <= ref_t[0]
    int idx = (int)(std::upper_bound(imu.begin(), imu.end(), ref_t[0]) - imu.begin());
    int64_t latest = ref_t[0];
    for (int r = 1; r < n_ref; r++) {
        const int64_t rts = ref_t[r];
        while (idx < n && imu[idx] <= rts) {
            if (imu[idx] > latest && idx > 0) { W.steps.push_back({idx, imu[idx] - latest}); W.refIdx.push_back(r); W.refCnt[r]++; }
            latest = imu[idx];
            ++idx;
        }
        if (idx > 0 && idx < n && rts > latest) { W.steps.push_back({idx, rts - latest}); W.refIdx.push_back(r); W.refCnt[r]++; }
        latest = rts;
    }
}

// ---- device: AccelerometerCalibrator::eval on the prepared streams, LBFGSSolver::minimize ----
//
// One workgroup of FIVE waves per window.  A loss evaluation walks ~10^4 steps whose sums must stay in
// time order, but only the sums are sequential: per step the expensive part (rotate the bias-corrected
// acceleration into the fixed frame, scale by dt; dt*v; the nine gradient products) is a function of
// the parameters and that step's data, and the gradient sums of a GPS interval need nothing from the
// forward sums but that interval's d_loss_d_travel.  So the window's chunks of <= 128 steps flow through
// a pipeline, one workgroup barrier per chunk ("tick"); every stage is either a lane = step map (P*)
// or ONE running fp64 sum per lane:
//   PF  (lane = step)  tick t:   a*dt of chunk t            -> LDS;   dt*v of chunk t-2  -> LDS
//   V   (lanes x,y,z)  tick t:   v += a*dt over chunk t-1, every v_i -> LDS
//   T   (lanes x,y,z)  tick t:   travel += dt*v over chunk t-3; at the end of a GPS interval hands travel over
//   B   (9 lanes)      tick t:   the interval ends F delivered one tick earlier: loss term, d_loss_d_travel
//                                (one divide per lane);  the 9 gradient sums over chunk t-lag
//   PB  (lane = step)  tick t:   the 9 gradient products of chunk t-lag+1 -> LDS
// lag = (most chunks in one interval) + 5 ticks, odd, make every hand-over visible in time.  The streams
// are read two chunks ahead into registers.  A tick costs what the slowest stage costs -- a running sum
// of 64 steps at ~13 cycles each (one dependent fp64 add plus its operand's LDS read: an LDS instruction
// costs the wave ~5 issue cycles whatever the number of active lanes, tools/ubench/fp64_chain.hip) --
// instead of the sum of all stages.
// History: one LANE per window with the whole loop body serial needed 68 s for a one-hour ride (every
// step waited for HBM); one WAVE per window doing the stages one after the other 2.5 s; three waves
// (producer, forward sums, backward sums) 1.5 s.  The fit of the slowest window bounds the run, so the
// latency of ONE evaluation is what counts; the machine is otherwise nearly idle (0.1 s of its time).

#define CB_CHUNK 128                           // steps per tick: two 64-lane halves for the lane = step stages
#define CB_HALVES (CB_CHUNK / 64)
#define CB_NBITS 10                            // chunk metadata word: steps | interval << CB_NBITS | last-of-interval << 31
#define CB_NMASK ((1 << CB_NBITS) - 1)
#define CB_RMASK ((1 << (31 - CB_NBITS)) - 1)
#define CB_BATCH 16
#define CB_ROWS (CB_CHUNK + 2 * CB_BATCH)      // LDS rows per chunk buffer: the one-batch-ahead reads may run past the chunk
#define CB_WAVES 5
#define CB_LAG_EXTRA 5

struct WinDesc {              // one per workgroup (window)
    int64_t fwd, bwd;         // element offsets of the SoA streams: fwd[CB_FWD][S], bwd[CB_BWD][S]
    int64_t refDist;          // [nRef]
    int64_t meta;             // int32 [2 * (nChunks + lag + 4)]: per chunk {first step, n | interval << CB_NBITS | last-of-interval << 31}
    int32_t S, nRef, nChunks, lag;
    double totalSec;
};

struct FwdRegs { double dt, qw, qx, qy, qz, ax, ay, az; };
struct BwdRegs { double c1, dt, m[9]; };

__device__ inline FwdRegs load_fwd(const double* F, int S, int idx)
{
    return {F[idx], F[S + idx], F[2 * S + idx], F[3 * S + idx], F[4 * S + idx], F[5 * S + idx], F[6 * S + idx], F[7 * S + idx]};
}
__device__ inline BwdRegs load_bwd(const double* B, int S, int idx)
{
    BwdRegs b;
    b.c1 = B[idx]; b.dt = B[S + idx];
    for (int k = 0; k < 9; k++) b.m[k] = B[(2 + k) * S + idx];
    return b;
}

__device__ inline double dot9(const double* a, const double* b)       // Eigen SSE2 redux order (contract E2)
{
    double l0 = a[0] * b[0], l1 = a[1] * b[1], m0 = a[2] * b[2], m1 = a[3] * b[3];
    l0 = l0 + a[4] * b[4]; l1 = l1 + a[5] * b[5];
    m0 = m0 + a[6] * b[6]; m1 = m1 + a[7] * b[7];
    l0 = l0 + m0; l1 = l1 + m1;
    return (l0 + l1) + a[8] * b[8];
}

struct WinView {
    const double *F, *B, *refDist;
    const int32_t* lmeta;         // LDS copy of the chunk list
    int S, nRef, nChunks, lag;
    double totalSec;
    // LDS; chunk buffers step-major so that the few lanes of a sum read neighbouring words (a field-major
    // layout put the 9 gradient rows one bank apart: 9-way conflicts)
    double (*sc)[CB_ROWS][5];     // [4][rows][5]: a*dt (x, y, z), dt of each step, pad (odd row pitch: a pitch of 4 doubles
                                  //                puts every 4th lane of a store on the same banks)     PF -> V, PF
    double (*sv)[CB_ROWS][3];     // [2][rows][3]: v after each step                        V -> PF
    double (*sm)[CB_ROWS][3];     // [2][rows][3]: dt*v of each step                        PF -> T
    double (*sp)[CB_ROWS][9];     // [2][rows][9]: the gradient products of each step       PB -> B
    double (*travel)[3];          // [nRef]: integrated_travel of each interval             T -> B
    double (*dvec)[3];            // [nRef]: d_loss_d_travel of each interval               B -> PB
    double* xch;                  // [10]: loss and gradient                                B -> everybody
};

// ---- the running sums of one chunk ----
// An LDS read takes ~100 cycles and the compiler keeps LDS reads next to their use, so the operands of
// a batch of steps are read into registers one batch ahead (the empty asm keeps the reads above the
// arithmetic of the previous batch).  The batch is sized so that a wave never has more than 15 LDS
// operations in flight: lgkmcnt is a 4-bit counter and the 16th operation stalls the wave until the
// first one has come back -- with 16-step batches of two-operand or read + write chains that stall,
// not the arithmetic, set the pace (34 cycles per step instead of 11).
template <int BATCH> struct Batch { double p[BATCH]; };

template <int STRIDE, int BATCH>
__device__ inline void batch_read(const double* src, int i0, Batch<BATCH>& b)
{
#pragma unroll
    for (int j = 0; j < BATCH; j++) b.p[j] = src[(i0 + j) * STRIDE];
}

// acc += src[i * STRIDE] for i = 0 .. n-1, in order; with OUT also out[i * 3] = acc after every step
template <int STRIDE, bool OUT, int BATCH>
__device__ inline void sum_chain(const double* src, int n, double& acc, double* out)
{
    double a = acc;
    int i = 0;
    auto steps = [&](const Batch<BATCH>& b, int i0) {
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            a = a + b.p[j];
            if (OUT) out[(i0 + j) * 3] = a;
        }
    };
    if (n >= BATCH) {                          // whole batches (rows past the chunk are padding)
        Batch<BATCH> A, B;
        batch_read<STRIDE, BATCH>(src, 0, A);
        batch_read<STRIDE, BATCH>(src, BATCH, B);
        asm volatile("" ::: "memory");
        steps(A, 0);
        i = BATCH;
        while (i + BATCH <= n) {
            batch_read<STRIDE, BATCH>(src, i + BATCH, A);
            asm volatile("" ::: "memory");
            steps(B, i);
            i += BATCH;
            if (i + BATCH > n) break;
            batch_read<STRIDE, BATCH>(src, i + BATCH, B);
            asm volatile("" ::: "memory");
            steps(A, i);
            i += BATCH;
        }
    }
    for (; i < n; i++) {                       // the odd steps of a short chunk
        a = a + src[i * STRIDE];
        if (OUT) out[i * 3] = a;
    }
    acc = a;
}

// The tick barrier.  __syncthreads() would do, but the compiler puts "wait for ALL memory operations" in front
// of every s_barrier it knows about -- including the producer's stream loads for the next chunks, i.e. one trip
// to HBM per tick (that, not arithmetic, set the pace of the first versions: 1.1 us per chunk).  Only LDS traffic
// has to be ordered here: the streams are read-only.
__device__ inline void cb_tick_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// velocity.cc:42-180 on the prepared streams.  Every wave returns the same loss and gradient.
// All waves run the same tick loop with the same trip count; the barrier is outside the roles.
// Chunk metadata sits in LDS (copied once per kernel): scalar loads would share a counter with the LDS
// reads of the sums and, returning out of order, turn each of their waits into "wait for everything".
__device__ double cal_eval(const WinView& G, int wave, int lane, const double* x, double* grad)
{
    const int T = G.nChunks + G.lag;
    auto start_of = [&](int j) { return __builtin_amdgcn_readfirstlane(G.lmeta[2 * (j < 0 ? 0 : j)]); };
    auto word_of = [&](int j) { return __builtin_amdgcn_readfirstlane(G.lmeta[2 * (j < 0 ? 0 : j) + 1]); };
    auto in_range = [&](int j) { return j >= 0 && j < G.nChunks; };
    if (wave == 3) {                                      // ---- PF: a*dt of chunk t, dt*v of chunk t-2 ----
        const double bg[3] = {x[0], x[1], x[2]}, bl[3] = {x[3], x[4], x[5]};
        // The stream is read TWO chunks ahead (a tick is shorter than a trip to HBM): two register sets, the
        // tick loop unrolled by two so that each set keeps its registers.
        FwdRegs fA[CB_HALVES], fB[CB_HALVES];
        for (int h = 0; h < CB_HALVES; h++) { fA[h] = load_fwd(G.F, G.S, start_of(0) + 64 * h + lane); fB[h] = load_fwd(G.F, G.S, start_of(1) + 64 * h + lane); }
        auto tick = [&](int t, FwdRegs (&fr)[CB_HALVES]) {
            const int startF = start_of(t + 2);
            const bool on = in_range(t);
            const int jm = t - 2;
            const bool onM = in_range(jm);
#pragma unroll
            for (int h = 0; h < CB_HALVES; h++) {
                const int row = 64 * h + lane;
                // the load is unconditional (past the end it re-reads the last chunk): a load under an `if` makes the
                // register set a phi, and the copies at the join wait for the loads just issued
                const FwdRegs f = fr[h];
                fr[h] = load_fwd(G.F, G.S, startF + row);
                if (on) {                                 // IntegrateMotion's parameter-dependent part (geometry.cc:34-45)
                    const Quat q = {f.qw, f.qx, f.qy, f.qz};
                    const double lc[3] = {f.ax + bl[0], f.ay + bl[1], f.az + bl[2]};
                    double rot[3];
                    quat_rotate(q, lc, rot);
                    double (*o)[5] = G.sc[t & 3];
                    for (int k = 0; k < 3; k++) o[row][k] = (rot[k] + bg[k]) * f.dt;
                    o[row][3] = f.dt;
                }
                if (onM) {                                // the product of `travel += dt * v` (velocity.cc:105-106)
                    const double dt = G.sc[jm & 3][row][3];
                    for (int k = 0; k < 3; k++) G.sm[jm & 1][row][k] = dt * G.sv[jm & 1][row][k];
                }
            }
            cb_tick_barrier();
        };
        for (int t = 0; t < T; t += 2) {
            tick(t, fA);
            if (t + 1 < T) tick(t + 1, fB);
        }
    } else if (wave == 4) {                               // ---- PB: gradient products of chunk t-lag+1 (velocity.cc:133-163) ----
        // lag is odd, so backward chunk jb = t - lag + 1 has the parity of t: the same two-set scheme
        BwdRegs bA[CB_HALVES], bB[CB_HALVES];
        for (int h = 0; h < CB_HALVES; h++) { bA[h] = load_bwd(G.B, G.S, start_of(0) + 64 * h + lane); bB[h] = load_bwd(G.B, G.S, start_of(1) + 64 * h + lane); }
        auto tick = [&](int t, BwdRegs (&br)[CB_HALVES]) {
            const int jb = t - G.lag + 1;
            const int startB = start_of(jb + 2), wordB = word_of(jb);
            const bool on = in_range(jb);
            const int r = (wordB >> CB_NBITS) & CB_RMASK;
            double d[3] = {0, 0, 0};
            if (on) { d[0] = G.dvec[r][0]; d[1] = G.dvec[r][1]; d[2] = G.dvec[r][2]; }
#pragma unroll
            for (int h = 0; h < CB_HALVES; h++) {
                const int row = 64 * h + lane;
                const BwdRegs b = br[h];
                br[h] = load_bwd(G.B, G.S, startB + row);
                if (on) {
                    double (*o)[9] = G.sp[jb & 1];
                    for (int k = 0; k < 3; k++) {
                        o[row][k] = b.c1 * d[k];
                        const double rw[3] = {b.m[3 * k], b.m[3 * k + 1], b.m[3 * k + 2]};
                        o[row][3 + k] = dot3(rw, d);
                        o[row][6 + k] = b.dt * d[k];
                    }
                }
            }
            cb_tick_barrier();
        };
        for (int t = 0; t < T; t += 2) {
            tick(t, bA);
            if (t + 1 < T) tick(t + 1, bB);
        }
    } else if (wave == 0) {                               // ---- V: v += a*dt (velocity.cc:99-101), every v_i kept ----
        const int comp = lane % 3;
        double vk = x[6 + comp];
        for (int t = 0; t < T; t++) {
            const int j = t - 1;
            // three lanes only: 64 lanes storing to three words would be serialised by the LDS as 21-way conflicts
            if (in_range(j) && lane < 3) sum_chain<5, true, 8>(&G.sc[j & 3][0][comp], word_of(j) & CB_NMASK, vk, &G.sv[j & 1][0][comp]);
            cb_tick_barrier();
        }
    } else if (wave == 1) {                               // ---- T: travel += dt*v (velocity.cc:105-108) ----
        const int comp = lane % 3;
        double tk = 0;
        for (int t = 0; t < T; t++) {
            const int j = t - 3;
            if (in_range(j)) {
                const int m = word_of(j);
                if (lane < 3) sum_chain<3, false, 16>(&G.sm[j & 1][0][comp], m & CB_NMASK, tk, nullptr);
                if (m < 0) {                              // last chunk of its interval
                    if (lane < 3) G.travel[(m >> CB_NBITS) & CB_RMASK][lane] = tk;
                    tk = 0;
                }
            }
            cb_tick_barrier();
        }
    } else {                                              // ---- B: interval ends and the gradient sums ----
        const int acc = lane % 9, comp = lane % 3;
        double gj = 0, result = 0;
        for (int t = 0; t < T; t++) {
            const int je = t - 4, j = t - G.lag;
            const int we = word_of(je), wj = word_of(j);
            if (in_range(je) && we < 0) {                 // T finished an interval during the previous tick (velocity.cc:118-127)
                const int r = (we >> CB_NBITS) & CB_RMASK;
                const double travel[3] = {G.travel[r][0], G.travel[r][1], G.travel[r][2]};
                const double tn = sqrt(dot3(travel, travel));
                const double diff = tn - G.refDist[r];
                result += diff * diff;
                if (lane < 3) G.dvec[r][lane] = ((2.0 * diff) * travel[comp]) / (tn + 1e-5);
            }
            if (in_range(j) && lane < 9) sum_chain<9, false, 16>(&G.sp[j & 1][0][acc], wj & CB_NMASK, gj, nullptr);
            cb_tick_barrier();
        }
        if (lane < 9) G.xch[1 + lane] = gj;
        if (lane == 0) G.xch[0] = result;
    }
    __syncthreads();
    const double fx = G.xch[0] / G.totalSec;
    for (int k = 0; k < 9; k++) grad[k] = G.xch[1 + k] / G.totalSec;
    __syncthreads();                                      // xch is rewritten only at the end of the next evaluation, but keep the waves together
    return fx;
}

// LBFGS.h:78-181 with Backtracking/Armijo (LineSearch.h:41-109); n = 9, m = 6, ftol 1e-4, 20 trials.
// Every lane of every wave carries the same solver state (the evaluations return identical doubles to all),
// so all three waves take the same branches and meet at the same barriers.
__device__ int cal_lbfgs(const WinView& G, int wave, int lane, double* x, double* fx_out, double epsilon, int max_iterations)
{
    double s[CB_M][9], y[CB_M][9], ysh[CB_M], alpha[CB_M], xp[9], grad[9], gradp[9], drt[9];
    double fx = cal_eval(G, wave, lane, x, grad);
    double xnorm = sqrt(dot9(x, x)), gnorm = sqrt(dot9(grad, grad));
    if (gnorm <= epsilon * fmax(xnorm, 1.0)) { *fx_out = fx; return 1; }
    for (int i = 0; i < 9; i++) drt[i] = -grad[i];
    double step = 1.0 / sqrt(dot9(drt, drt));
    int k = 1, end = 0;
    for (;;) {
        for (int i = 0; i < 9; i++) { xp[i] = x[i]; gradp[i] = grad[i]; }
        const double fx_init = fx, dg_init = dot9(grad, drt), dg_test = 1e-4 * dg_init;
        for (int iter = 0; iter < 20; iter++) {
            for (int i = 0; i < 9; i++) x[i] = xp[i] + step * drt[i];
            fx = cal_eval(G, wave, lane, x, grad);
            if (!(fx > fx_init + step * dg_test)) break;
            if (step < 1e-20) { *fx_out = fx; return -2; }        // the reference throws here
            if (step > 1e+20) { *fx_out = fx; return -3; }
            step *= 0.5;
        }
        xnorm = sqrt(dot9(x, x)); gnorm = sqrt(dot9(grad, grad));
        if (gnorm <= epsilon * fmax(xnorm, 1.0)) { *fx_out = fx; return k; }
        if (max_iterations != 0 && k >= max_iterations) { *fx_out = fx; return k; }
        for (int i = 0; i < 9; i++) { s[end][i] = x[i] - xp[i]; y[end][i] = grad[i] - gradp[i]; }
        const double ys = dot9(y[end], s[end]), yy = dot9(y[end], y[end]);
        ysh[end] = ys;
        for (int i = 0; i < 9; i++) drt[i] = -grad[i];
        const int bound = CB_M < k ? CB_M : k;
        end = (end + 1) % CB_M;
        int j = end;
        for (int t = 0; t < bound; t++) {
            j = (j + CB_M - 1) % CB_M;
            alpha[j] = dot9(s[j], drt) / ysh[j];
            for (int i = 0; i < 9; i++) drt[i] -= alpha[j] * y[j][i];
        }
        const double sc = ys / yy;
        for (int i = 0; i < 9; i++) drt[i] *= sc;
        for (int t = 0; t < bound; t++) {
            const double beta = dot9(y[j], drt) / ysh[j];
            const double ab = alpha[j] - beta;
            for (int i = 0; i < 9; i++) drt[i] += ab * s[j][i];
            j = (j + 1) % CB_M;
        }
        step = 1.0;
        k++;
    }
}

// mode 0: L-BFGS from x = 0 (fit_motion.cc:186-190).  mode 1: one evaluation at xin (tests).
__global__ __launch_bounds__(64 * CB_WAVES) void k_calibrate_windows(const WinDesc* __restrict__ wins, const double* __restrict__ dbl,
                                                                   const int32_t* __restrict__ i32, int mode, const double* __restrict__ xin,
                                                                   int max_iterations, double* __restrict__ xout, double* __restrict__ fxout,
                                                                   double* __restrict__ gradout, int32_t* __restrict__ niter)
{
    extern __shared__ __attribute__((aligned(16))) double cb_lds[];
    const WinDesc D = wins[blockIdx.x];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x % 64;
    const size_t w = blockIdx.x;
    WinView G;
    G.F = dbl + D.fwd; G.B = dbl + D.bwd; G.refDist = dbl + D.refDist;
    G.S = D.S; G.nRef = D.nRef; G.nChunks = D.nChunks; G.lag = D.lag; G.totalSec = D.totalSec;
    double* l = cb_lds;
    G.sc = (double (*)[CB_ROWS][5])l;      l += 4 * CB_ROWS * 5;
    G.sv = (double (*)[CB_ROWS][3])l;      l += 2 * CB_ROWS * 3;
    G.sm = (double (*)[CB_ROWS][3])l;      l += 2 * CB_ROWS * 3;
    G.sp = (double (*)[CB_ROWS][9])l;      l += 2 * CB_ROWS * 9;
    G.xch = l;                             l += 16;
    G.travel = (double (*)[3])l;           l += 3 * (size_t)D.nRef;
    G.dvec = (double (*)[3])l;             l += 3 * (size_t)D.nRef;
    int32_t* lm = (int32_t*)l;
    for (int i = threadIdx.x; i < 2 * (D.nChunks + D.lag + 4); i += 64 * CB_WAVES) lm[i] = i32[D.meta + i];
    G.lmeta = lm;
    // Roles by placement: five waves on four SIMDs means one SIMD carries two.  The two running sums that wait
    // most (T, B: one dependent add per step) share it; V, which sets the pace, and the two producers get a SIMD
    // each.  HW_ID[5:4] is the SIMD the wave landed on.
    int32_t* simdOf = lm + 2 * (D.nChunks + D.lag + 4);
    if (lane == 0) simdOf[wave] = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));
    __syncthreads();
    int role;
    {
        int cnt[4] = {0, 0, 0, 0}, shared = 0;
        for (int i = 0; i < CB_WAVES; i++) cnt[simdOf[i] & 3]++;
        while (shared < 3 && cnt[shared] < 2) shared++;
        int pair = 0, single = 0;
        role = 0;
        for (int i = 0; i < CB_WAVES; i++) {
            const bool inPair = (simdOf[i] & 3) == shared && pair < 2;
            const int r = inPair ? 1 + pair : (single == 0 ? 0 : 2 + single);      // pair -> T (1), B (2); singles -> V (0), PF (3), PB (4)
            if (inPair) pair++; else single++;
            if (i == wave) role = r;
        }
        role = __builtin_amdgcn_readfirstlane(role);
    }
    double x[9], g[9], fx;
    if (mode == 1) {
        for (int i = 0; i < 9; i++) x[i] = xin[9 * w + i];
        fx = cal_eval(G, role, lane, x, g);
        if (wave == 0 && lane < 9) gradout[9 * w + lane] = g[lane];
        if (wave == 0 && lane == 0) fxout[w] = fx;
        return;
    }
    for (int i = 0; i < 9; i++) x[i] = 0.0;
    const int it = cal_lbfgs(G, role, lane, x, &fx, 1e-5, max_iterations);
    if (wave == 0 && lane < 9) xout[9 * w + lane] = x[lane];
    if (wave == 0 && lane == 0) { fxout[w] = fx; niter[w] = it; }
}

// ---- host: pack, launch ----

struct Packed {
    std::vector<WinDesc> wins;
    std::unique_ptr<double[]> dbl;       // not value-initialised: every element is written by fill_window (864 MB for a one-hour ride)
    size_t dblCount = 0;
    std::vector<int32_t> i32;
};

// Sizes and the chunk list of one window (cheap, serial); the streams are filled in by fill_window.
void layout_window(const Window& W, Packed& P)
{
    WinDesc D;
    const size_t n = W.steps.size();
    D.S = (int32_t)((n + CB_CHUNK - 1) / CB_CHUNK * CB_CHUNK + CB_CHUNK);   // a chunk load may start at the last step
    D.nRef = (int32_t)W.refCnt.size();
    D.totalSec = 0;
    D.fwd = (int64_t)P.dblCount;     P.dblCount += (size_t)D.S * CB_FWD;
    D.bwd = (int64_t)P.dblCount;     P.dblCount += (size_t)D.S * CB_BWD;
    D.refDist = (int64_t)P.dblCount; P.dblCount += (size_t)D.nRef;
    // chunk list: {first step, n | interval << CB_NBITS | last-of-interval << 31}; intervals without steps have no chunk
    // (their loss term and gradient contribution are exactly +0: travel = 0, reference_distance = 0)
    D.meta = (int64_t)P.i32.size();
    int32_t off = 0, last = 0, nchunks = 0, maxPer = 0;
    for (size_t r = 0; r < W.refCnt.size(); r++) {
        const int32_t c = W.refCnt[r];
        int32_t per = 0;
        for (int32_t c0 = 0; c0 < c; c0 += CB_CHUNK, per++, nchunks++) {
            const int32_t nn = std::min(CB_CHUNK, c - c0);
            last = off + c0;
            P.i32.push_back(last);
            P.i32.push_back(nn | ((int32_t)r << CB_NBITS) | (c0 + CB_CHUNK >= c ? (int32_t)0x80000000 : 0));
        }
        maxPer = std::max(maxPer, per);
        off += c;
    }
    D.nChunks = nchunks;
    D.lag = (maxPer + CB_LAG_EXTRA) | 1;                                                     // odd: see the producers
    for (int i = 0; i < D.lag + 4; i++) { P.i32.push_back(last); P.i32.push_back(0); }      // what the prefetcher reads past the end
    P.wins.push_back(D);
}

// Streams of one window as structure-of-arrays (a chunk of 64 consecutive steps of one field = 512 contiguous
// bytes), in the order of velocity.cc:62-168: both loops of a reference interval walk the same steps, and the
// quantities below depend on the data only.
void fill_window(const Window& W, const Imu& M, const double* ref_v, WinDesc& D, double* dbl)
{
    double* F = dbl + D.fwd; double* B = dbl + D.bwd; double* refDist = dbl + D.refDist;
    const size_t S = (size_t)D.S, n = W.steps.size();
    for (int r = 0; r < D.nRef; r++) refDist[r] = 0.0;
    for (size_t i = n; i < S; i++) {                       // padding rows the prefetcher may read
        for (int f = 0; f < CB_FWD; f++) F[f * S + i] = 0.0;
        for (int f = 0; f < CB_BWD; f++) B[f * S + i] = 0.0;
    }
    Quat q = {1.0, 0.0, 0.0, 0.0};
    double twr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int64_t total_usec = 0;
    for (size_t i = 0; i < n; i++) {
        const Step& St = W.steps[i];
        const double* rr = M.rot + 3 * (size_t)M.evRot[St.event];
        const double* aa = M.acc + 3 * (size_t)M.evAcc[St.event];
        const double dt = (double)St.usec * 1e-6;
        F[i] = dt; F[S + i] = q.w; F[2 * S + i] = q.x; F[3 * S + i] = q.y; F[4 * S + i] = q.z;
        F[5 * S + i] = aa[0]; F[6 * S + i] = aa[1]; F[7 * S + i] = aa[2];
        refDist[W.refIdx[i]] += dt * ref_v[W.refIdx[i]];
        q = quat_mul(q, rotation_motion_to_quaternion(rr[0], rr[1], rr[2], dt));
        total_usec += St.usec;
        const double total_sec = (double)total_usec * 1e-6;
        double R[9];
        quat_matrix(q, R);
        for (int k = 0; k < 9; k++) twr[k] += R[k] * dt;
        B[i] = total_sec * dt; B[S + i] = dt;
        for (int k = 0; k < 3; k++) {
            B[(2 + 3 * k) * S + i] = dt * twr[k]; B[(3 + 3 * k) * S + i] = dt * twr[3 + k]; B[(4 + 3 * k) * S + i] = dt * twr[6 + k];
        }
    }
    D.totalSec = (double)total_usec * 1e-6;
}

void pack_windows(const std::vector<Window>& wins, const Imu& M, const std::vector<const double*>& ref_v, Packed& P)
{
    for (const Window& W : wins) layout_window(W, P);
    P.dbl.reset(new double[P.dblCount ? P.dblCount : 1]);
    // the windows are independent: one host thread per core, windows handed out by an atomic counter
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (size_t w = next++; w < wins.size(); w = next++) fill_window(wins[w], M, ref_v[w], P.wins[w], P.dbl.get());
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthreads = std::min<size_t>(wins.size(), std::max(1u, std::min(hw ? hw : 1u, 32u)));
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nthreads; t++) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool put(const void* src, size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess && (!bytes || hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) == hipSuccess); }
    bool make(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess; }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool timing_on() { static const bool v = getenv("PGORB_CALIB_TIMING") != nullptr; return v; }

int run_windows(pgorb_ctx* c, const Packed& P, int mode, const double* xin, int max_iters, double* x, double* fx, double* grad, int32_t* niter)
{
    const double t0 = now_s();
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    const size_t nw = P.wins.size();
    DevBuf dW, dD, dI, dXin, dX, dF, dGr, dNi;
    if (!dW.put(P.wins.data(), nw * sizeof(WinDesc)) || !dD.put(P.dbl.get(), P.dblCount * 8) || !dI.put(P.i32.data(), P.i32.size() * 4) ||
        !dXin.put(xin, xin ? nw * 72 : 0) || !dX.make(nw * 72) || !dF.make(nw * 8) || !dGr.make(nw * 72) || !dNi.make(nw * 4))
        return pg_ctx_fail(c, PGORB_E_HIP, "device allocation / upload for the calibration windows failed");
    const double t1 = now_s();
    size_t maxRef = 1, maxMeta = 1;
    for (const WinDesc& D : P.wins) { maxRef = std::max(maxRef, (size_t)D.nRef); maxMeta = std::max(maxMeta, (size_t)(D.nChunks + D.lag + 4)); }
    const size_t lds = (CB_ROWS * (4 * 5 + 2 * 3 + 2 * 3 + 2 * 9) + 16 + 6 * maxRef + maxMeta + 4) * sizeof(double);
    if (lds > 160 * 1024) return pg_ctx_fail(c, PGORB_E_LIMIT, "locations_batch_size too large for the calibration workgroup's LDS");
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_calibrate_windows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_calibrate_windows, dim3((unsigned)nw), dim3(64 * CB_WAVES), lds, 0, (const WinDesc*)dW.p, (const double*)dD.p, (const int32_t*)dI.p,
                       mode, (const double*)dXin.p, max_iters, (double*)dX.p, (double*)dF.p, (double*)dGr.p, (int32_t*)dNi.p);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_calibrate_windows failed");
    if (timing_on())
        fprintf(stderr, "[calib] %zu windows, %.1f MB of streams: upload %.3f s, solver kernel %.1f us\n", nw, P.dblCount * 8e-6, t1 - t0, (now_s() - t1) * 1e6);
    bool ok = hipMemcpy(fx, dF.p, sizeof(double) * nw, hipMemcpyDeviceToHost) == hipSuccess;
    if (mode == 1) ok = ok && hipMemcpy(grad, dGr.p, sizeof(double) * 9 * nw, hipMemcpyDeviceToHost) == hipSuccess;
    else ok = ok && hipMemcpy(x, dX.p, sizeof(double) * 9 * nw, hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(niter, dNi.p, sizeof(int32_t) * nw, hipMemcpyDeviceToHost) == hipSuccess;
    return ok ? PGORB_OK : pg_ctx_fail(c, PGORB_E_HIP, "download of the calibration results failed");
}

bool build_windows(const int64_t* gps_t, int n_gps, const Imu& M, int batch, int shift, std::vector<Window>& wins, std::vector<int>& starts)
{
    for (int i = 0; i + 1 < n_gps; i++) if (!(gps_t[i] < gps_t[i + 1])) return false;
    for (int start = 0; start < n_gps; start += shift) {
        const int end = std::min(start + batch, n_gps);
        wins.emplace_back();
        make_window(gps_t + start, end - start, M, wins.back());
        starts.push_back(start);
    }
    return true;
}

}  // namespace

extern "C" {

int pgorb_fit_num_windows(int n_gps, int locations_shift_step)
{
    if (n_gps <= 0 || locations_shift_step <= 0) return 0;
    return (n_gps + locations_shift_step - 1) / locations_shift_step;
}

int pgorb_fit_velocity_windows(pgorb_ctx* c, const double* gps_velocity, const int64_t* gps_time_usec, int n_gps,
                               const double* rotations, const int64_t* rot_time_usec, int n_rot,
                               const double* accelerations, const int64_t* acc_time_usec, int n_acc,
                               int locations_batch_size, int locations_shift_step, int optimization_iters,
                               double* x, double* residual, int32_t* niter)
{
    if (!c) return PGORB_E_ARG;
    if (!gps_velocity || !gps_time_usec || !rotations || !rot_time_usec || !accelerations || !acc_time_usec || !x || !residual || !niter ||
        n_gps <= 0 || locations_batch_size <= 0 || locations_shift_step <= 0 || optimization_iters <= 0 ||
        locations_batch_size < locations_shift_step)                                   // the CHECKs of fit_motion.cc:300-306
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_fit_velocity_windows");
    Imu M;
    if (!merge_imu(rotations, rot_time_usec, n_rot, accelerations, acc_time_usec, n_acc, M))
        return pg_ctx_fail(c, PGORB_E_ARG, "rotation / acceleration time series cannot be merged (empty, unordered or disjoint)");
    std::vector<Window> wins; std::vector<int> starts;
    if (!build_windows(gps_time_usec, n_gps, M, locations_batch_size, locations_shift_step, wins, starts))
        return pg_ctx_fail(c, PGORB_E_ARG, "GPS timestamps must increase");
    std::vector<const double*> refv;
    for (int s : starts) refv.push_back(gps_velocity + s);
    Packed P;
    const double t0 = now_s();
    pack_windows(wins, M, refv, P);
    if (timing_on()) fprintf(stderr, "[calib] host preparation of the streams %.3f s\n", now_s() - t0);
    return run_windows(c, P, 0, nullptr, optimization_iters, x, residual, nullptr, niter);
}

int pgorb_calibrator_eval(pgorb_ctx* c, const double* gps_velocity, const int64_t* gps_time_usec, int n_gps,
                          const double* rotations, const int64_t* rot_time_usec, int n_rot,
                          const double* accelerations, const int64_t* acc_time_usec, int n_acc,
                          const double* xin, int n_points, double* fx, double* grad)
{
    if (!c) return PGORB_E_ARG;
    if (!gps_velocity || !gps_time_usec || !rotations || !rot_time_usec || !accelerations || !acc_time_usec || !xin || !fx || !grad ||
        n_gps <= 0 || n_points <= 0)
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_calibrator_eval");
    Imu M;
    if (!merge_imu(rotations, rot_time_usec, n_rot, accelerations, acc_time_usec, n_acc, M))
        return pg_ctx_fail(c, PGORB_E_ARG, "rotation / acceleration time series cannot be merged (empty, unordered or disjoint)");
    for (int i = 0; i + 1 < n_gps; i++) if (!(gps_time_usec[i] < gps_time_usec[i + 1])) return pg_ctx_fail(c, PGORB_E_ARG, "GPS timestamps must increase");
    std::vector<Window> wins(1);
    make_window(gps_time_usec, n_gps, M, wins[0]);
    Packed P;
    pack_windows(wins, M, std::vector<const double*>(1, gps_velocity), P);
    P.wins.resize(n_points, P.wins[0]);                      // the same calibrator (shared streams) at n_points parameter vectors
    return run_windows(c, P, 1, xin, 0, nullptr, fx, grad, nullptr);
}

int pgorb_fit_motion_velocities(pgorb_ctx* c, const double* gps_velocity, const int64_t* gps_time_usec, int n_gps,
                                const double* rotations, const int64_t* rot_time_usec, int n_rot,
                                const double* accelerations, const int64_t* acc_time_usec, int n_acc,
                                const double* vertical_axis, int locations_batch_size, int locations_shift_step,
                                int optimization_iters, double post_smoothing_sigma_sec,
                                double forward_axis_inference_min_velocity_m_s, double forward_axis_inference_min_rotation_rad,
                                int64_t* out_time_usec, double* out_velocity, int* n_out, double* forward_axis)
{
    if (!c) return PGORB_E_ARG;
    if (!vertical_axis || !out_time_usec || !out_velocity || !n_out || !forward_axis || !(post_smoothing_sigma_sec > 0))
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_fit_motion_velocities");
    const int nw = pgorb_fit_num_windows(n_gps, locations_shift_step);
    std::vector<double> x((size_t)std::max(nw, 1) * 9), res(std::max(nw, 1));
    std::vector<int32_t> it(std::max(nw, 1));
    int rc = pgorb_fit_velocity_windows(c, gps_velocity, gps_time_usec, n_gps, rotations, rot_time_usec, n_rot, accelerations, acc_time_usec,
                                        n_acc, locations_batch_size, locations_shift_step, optimization_iters, x.data(), res.data(), it.data());
    if (rc) return rc;
    for (int w = 0; w < nw; w++) if (it[w] < 0) return pg_ctx_fail(c, PGORB_E_LIMIT, "the line search step left [1e-20, 1e20] (LBFGSpp throws here)");
    // fit_motion.cc:196-246 with the fitted parameters: IntegrateTrajectory (velocity.cc:200-253) per window.  The
    // windows are integrated on all host cores; what the reference accumulates across windows (the per-sample lists
    // it later sums front to back, the Kahan sum of local-frame velocities) is then merged in window order.
    Imu M;
    merge_imu(rotations, rot_time_usec, n_rot, accelerations, acc_time_usec, n_acc, M);
    const int nev = (int)M.time.size();
    struct WinOut {
        std::vector<int32_t> ev;            // the merged events the window reaches, ascending (the keys of the std::map)
        std::vector<double> nrm;            // velocity.norm() per event
        std::vector<double> vl;             // local-frame velocity (3 per entry) of the events that count for the forward axis
    };
    std::vector<WinOut> outs((size_t)nw);
    {
        std::atomic<int> next(0);
        auto work = [&]() {
            std::vector<Quat> ori; std::vector<double> vel;
            for (int w = next++; w < nw; w = next++) {
                const int start = w * locations_shift_step, end = std::min(start + locations_batch_size, n_gps);
                Window W;
                make_window(gps_time_usec + start, end - start, M, W);
                WinOut& O = outs[w];
                ori.clear(); vel.clear();
                const double* p = &x[(size_t)w * 9];
                Quat q = {1.0, 0.0, 0.0, 0.0};
                double v[3] = {p[6], p[7], p[8]};
                for (const Step& S : W.steps) {
                    const double* rr = rotations + 3 * (size_t)M.evRot[S.event];
                    const double* aa = accelerations + 3 * (size_t)M.evAcc[S.event];
                    const double dt = (double)S.usec * 1e-6;
                    const double lc[3] = {aa[0] + p[3], aa[1] + p[4], aa[2] + p[5]};
                    double rot[3];
                    quat_rotate(q, lc, rot);
                    for (int k = 0; k < 3; k++) v[k] = v[k] + (rot[k] + p[k]) * dt;
                    q = quat_mul(q, rotation_motion_to_quaternion(rr[0], rr[1], rr[2], dt));
                    if (O.ev.empty() || O.ev.back() != S.event) { O.ev.push_back(S.event); ori.push_back(q); vel.insert(vel.end(), v, v + 3); }
                    else { ori.back() = q; std::copy(v, v + 3, vel.end() - 3); }   // an interval split at a GPS fix: the later part wins (:244-248)
                }
                double min_rotation_cos = 1.0;
                O.nrm.resize(O.ev.size());
                for (size_t i = 0; i < O.ev.size(); i++) {
                    O.nrm[i] = sqrt(dot3(&vel[3 * i], &vel[3 * i]));
                    min_rotation_cos = std::min(min_rotation_cos, std::abs(ori[i].w));
                }
                if (acos(min_rotation_cos) >= forward_axis_inference_min_rotation_rad)
                    for (size_t i = 0; i < O.ev.size(); i++) if (O.nrm[i] >= forward_axis_inference_min_velocity_m_s) {
                        const Quat inv = {ori[i].w, -ori[i].x, -ori[i].y, -ori[i].z};
                        double vl[3];
                        quat_rotate(inv, &vel[3 * i], vl);
                        O.vl.insert(O.vl.end(), vl, vl + 3);
                    }
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const int nthreads = std::max(1, std::min<int>(nw, std::min(hw ? hw : 1u, 32u)));
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; t++) pool.emplace_back(work);
        work();
        for (std::thread& t : pool) t.join();
    }
    std::vector<double> vsum(nev, 0.0);                                     // std::accumulate(list, 0.0) = the running sum in push order
    std::vector<int32_t> vcnt(nev, 0);
    double ksum[3] = {0, 0, 0}, krem[3] = {0, 0, 0};
    for (int w = 0; w < nw; w++) {
        const WinOut& O = outs[w];
        for (size_t i = 0; i < O.ev.size(); i++) { vsum[O.ev[i]] += O.nrm[i]; vcnt[O.ev[i]]++; }
        for (size_t i = 0; i + 2 < O.vl.size(); i += 3)
            for (int k = 0; k < 3; k++) {                                  // KahanSum::add (math.hpp:13-19)
                const double proposed = O.vl[i + k] + krem[k], updated = ksum[k] + proposed, actual = updated - ksum[k];
                krem[k] = proposed - actual; ksum[k] = updated;
            }
    }
    std::vector<double> avg, tsec;
    int n = 0;
    for (int e = 0; e < nev; e++) if (vcnt[e]) {
        out_time_usec[n] = M.time[e];
        tsec.push_back((double)(out_time_usec[n] - out_time_usec[0]) * 1e-6);
        avg.push_back(vsum[e] / vcnt[e]);
        n++;
    }
    if (n && pgorb_smooth_time_series(avg.data(), tsec.data(), n, tsec.data(), n, post_smoothing_sigma_sec, out_velocity) != PGORB_OK)
        return pg_ctx_fail(c, PGORB_E_ARG, "SmoothTimeSeries failed");
    *n_out = n;
    double f[3] = {ksum[0], ksum[1], ksum[2]};
    const double dp = dot3(vertical_axis, f);
    for (int k = 0; k < 3; k++) f[k] -= vertical_axis[k] * dp;
    const double nn = sqrt(dot3(f, f)) + 1e-5;
    for (int k = 0; k < 3; k++) forward_axis[k] = f[k] / nn;
    return PGORB_OK;
}

}  // extern "C"

// calib.hip -- K9: fit_motion's sliding-window accelerometer calibration, every window's L-BFGS
// fit running at once on the GPU (BASELINE.json configs[4]).
//
// Reference: for every window of 40 GPS fixes (step 5) fit_motion builds an AccelerometerCalibrator
// and minimises its 9-parameter loss with LBFGSpp (src/fit_motion.cc:151-246,
// src/calibration/velocity.cc:42-180, thirdparty/LBFGS/LBFGS.h:78-181, LBFGS/LineSearch.h:41-109);
// one loss evaluation integrates every IMU sample of the window in time order
// (src/geometry/geometry.cc:24-53), ~10^4 dependent steps, and the fits run one after another.
//
// Here: windows are independent, so ONE WAVE PER WINDOW runs the whole solver, all windows at once.
// Every double is produced by the same operation on the same operands as in the reference (the sums
// stay sequential in time: reordering them changes the doubles and the JSON is diffed bit for
// bit); inside a window the lanes share out what is independent per step (see the device section).
// What does not depend on the nine parameters is hoisted out of the solver and computed once per
// window on the host, with the same operations in the same order as the reference's loop body:
//   forward stream  (8 doubles / step): dt, orientation BEFORE the step (w,x,y,z), raw acceleration
//   backward stream (11 doubles / step): total_time_sec*dt, dt, dt * total_time_weighted_rotation^T (3x3)
//   per GPS interval: reference_distance, first step
// (RotationMotionToQuaternion's sin/cos therefore run in the host libm, as in the reference.)
// Streams are structure-of-arrays per window, so a chunk of 64 steps of one field is one coalesced
// 512-byte request.  fp64 add/mul/div/sqrt are IEEE on both sides, contraction is off (Makefile),
// so a wave reproduces the CPU run of its window bit for bit.
//
// Eigen reduction orders (the parity contract shared with oracle/calib_oracle.c, E0-E5 there):
// 3-vectors t0 + (t1 + t2); the solver's 9-vectors in SSE2 packet order.
#include "pgorb_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
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
// One wave per window.  A loss evaluation walks ~10^4 steps whose sums must stay in time order, but
// only the sums are sequential: per step the expensive part (rotate the bias-corrected acceleration
// into the fixed frame, scale by dt; the nine gradient products) depends on the parameters and on
// that step's data alone.  So for every chunk of 64 steps
//   phase A  lane = step: the independent arithmetic, results to LDS (full 64-lane efficiency);
//   phase S  lane = component: the running sums v, travel (3 lanes) / the 9 gradient sums (9 lanes)
//            walk the chunk in order, one dependent fp64 add per step and chain.
// The first design (one LANE per window, the whole loop body serial) spent ~1000 cycles per step and
// was 20x slower per evaluation than one CPU core; the fit of the slowest window bounds the whole
// run, so per-evaluation latency is what matters.  Chunks are prefetched one ahead into registers,
// across interval and pass boundaries (the streams do not depend on the parameters).

#define CB_CHUNK 64

// developer build: make EXTRA=-DPGORB_CALIB_PROF -- shader-clock cycles per phase of cal_eval, summed by lane 0 of block 0
#ifdef PGORB_CALIB_PROF
__device__ unsigned long long cb_prof[8];
#define CP_DECL unsigned long long cp_t = __builtin_readcyclecounter(), cp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CP_TICK(k) do { const unsigned long long cp_n = __builtin_readcyclecounter(); cp_acc[k] += cp_n - cp_t; cp_t = cp_n; } while (0)
#define CP_FLUSH do { if (lane == 0 && blockIdx.x == 0) for (int cp_i = 0; cp_i < 8; cp_i++) atomicAdd(&cb_prof[cp_i], cp_acc[cp_i]); } while (0)
#else
#define CP_DECL
#define CP_TICK(k)
#define CP_FLUSH
#endif

struct WinDesc {              // one per wave (window)
    int64_t fwd, bwd;         // element offsets of the SoA streams: fwd[CB_FWD][S], bwd[CB_BWD][S]
    int64_t refDist;          // [nRef]
    int64_t refOff;           // int32 [nRef + 1]: first step of each reference interval
    int64_t chunks;           // int32 [nChunks + 2]: first step of every chunk of <= 64 steps, in order (+ 2 repeats of the last)
    int32_t S, nRef;
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
    const int32_t *refOff, *chunks;
    int S, nRef;
    double totalSec;
    // LDS, step-major so that the few lanes of phase S read neighbouring words (a field-major layout put
    // the 9 rows one bank apart: 9-way conflicts, 23 cycles per dependent add instead of 8)
    double (*sc)[4];          // [64][4]: c0, c1, c2, dt of each step of the chunk
    double (*sp)[9];          // [64][9]: the gradient products of each step
};

// ---- phase S: the running sums of one chunk ----
// A dependent fp64 operation issues ~17 cycles after its producer and an LDS read takes ~100, and the
// compiler keeps LDS reads next to their use.  So (i) the operands of 16 steps are read into registers
// one batch ahead (the empty asm keeps the reads above the arithmetic of the previous batch), and
// (ii) the forward chains are skewed so that no operation of a slot waits for another one of the same
// slot.  travel + (+0.0) is exact (travel starts at +0.0 and a sum is -0.0 only if both terms are),
// which lets every slot have the same shape; the pending products are flushed, oldest first, at the
// end of the chunk.
#define CB_BATCH 16

struct FwdBatch { double c[CB_BATCH], dt[CB_BATCH]; };

__device__ inline void fwd_read(double (*sc)[4], int comp, int i0, FwdBatch& b)
{
#pragma unroll
    for (int j = 0; j < CB_BATCH; j++) { b.c[j] = sc[i0 + j][comp]; b.dt[j] = sc[i0 + j][3]; }
}

// One slot = { travel += m2 ; m1' = dt[i-1] * v ; v += c[i] }: the product is consumed two slots after it was
// issued (fp64 multiplies take about twice as long as adds here), so a slot only waits for the v chain.
// The scheduling barrier keeps the compiler from folding the slots back into product-then-add order.
template <bool FIRST>
__device__ inline void fwd_steps(const FwdBatch& b, double& v, double& m1, double& m2, double& dtp, double& t)
{
#pragma unroll
    for (int j = 0; j < CB_BATCH; j++) {
        if (!(FIRST && j == 0)) { t += m2; m2 = m1; m1 = dtp * v; }
        v = v + b.c[j];
        dtp = b.dt[j];
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ inline void fwd_chain(double (*sc)[4], int comp, int n, double& vk, double& tk)
{
    double v = vk, t = tk, m1 = 0.0, m2 = 0.0, dtp = 0.0;
    int i = 0;
    if (n >= CB_BATCH) {                       // whole batches, operands one batch ahead (rows past the chunk are padding)
        FwdBatch A, B;
        fwd_read(sc, comp, 0, A);
        fwd_read(sc, comp, CB_BATCH, B);
        asm volatile("" ::: "memory");
        fwd_steps<true>(A, v, m1, m2, dtp, t);
        i = CB_BATCH;
        while (i + CB_BATCH <= n) {
            fwd_read(sc, comp, i + CB_BATCH, A);
            asm volatile("" ::: "memory");
            fwd_steps<false>(B, v, m1, m2, dtp, t);
            i += CB_BATCH;
            if (i + CB_BATCH > n) break;
            fwd_read(sc, comp, i + CB_BATCH, B);
            asm volatile("" ::: "memory");
            fwd_steps<false>(A, v, m1, m2, dtp, t);
            i += CB_BATCH;
        }
    }
    for (; i < n; i++) {                       // the odd steps of a short chunk
        if (i > 0) { t += m2; m2 = m1; m1 = dtp * v; }
        v = v + sc[i][comp];
        dtp = sc[i][3];
    }
    t += m2;                                   // flush, oldest product first
    t += m1;
    t += dtp * v;
    vk = v; tk = t;
}

struct BwdBatch { double p[CB_BATCH]; };

__device__ inline void bwd_read(double (*sp)[9], int acc, int i0, BwdBatch& b)
{
#pragma unroll
    for (int j = 0; j < CB_BATCH; j++) b.p[j] = sp[i0 + j][acc];
}

__device__ inline void bwd_chain(double (*sp)[9], int acc, int n, double& gj)
{
    double g = gj;
    int i = 0;
    if (n >= CB_BATCH) {
        BwdBatch A, B;
        bwd_read(sp, acc, 0, A);
        bwd_read(sp, acc, CB_BATCH, B);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < CB_BATCH; j++) g += A.p[j];
        i = CB_BATCH;
        while (i + CB_BATCH <= n) {
            bwd_read(sp, acc, i + CB_BATCH, A);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < CB_BATCH; j++) g += B.p[j];
            i += CB_BATCH;
            if (i + CB_BATCH > n) break;
            bwd_read(sp, acc, i + CB_BATCH, B);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < CB_BATCH; j++) g += A.p[j];
            i += CB_BATCH;
        }
    }
    for (; i < n; i++) g += sp[i][acc];
    gj = g;
}

// velocity.cc:42-180 on the prepared streams; every lane returns the same loss and gradient.
__device__ double cal_eval(const WinView& G, int lane, const double* x, double* grad)
{
    const double bg[3] = {x[0], x[1], x[2]}, bl[3] = {x[3], x[4], x[5]};
    const int comp = lane % 3, acc = lane % 9;
    double vk = x[6 + comp];                          // this lane's component of integrated_velocity
    double gj = 0;                                    // this lane's gradient accumulator (index acc)
    double result = 0;
    CP_DECL;
    // chunk t+1 of each stream is loaded while chunk t is worked on; the chunk starts come from a flat list
    // (two entries ahead in SGPRs), so no address depends on a load of the same trip
    int fi = 0, bi = 0;
    FwdRegs fn = load_fwd(G.F, G.S, G.chunks[0] + lane);
    BwdRegs bn = load_bwd(G.B, G.S, G.chunks[0] + lane);
    int fnext = G.chunks[1], bnext = fnext;
    for (int r = 0; r < G.nRef; r++) {
        const int off = G.refOff[r], cnt = G.refOff[r + 1] - off;
        double tk = 0;                                // this lane's component of integrated_travel
        for (int c0 = 0; c0 < cnt; c0 += CB_CHUNK) {
            const int n = min(CB_CHUNK, cnt - c0);
            CP_TICK(0);
            const FwdRegs f = fn;
            fn = load_fwd(G.F, G.S, fnext + lane);
            fnext = G.chunks[++fi + 1];
            CP_TICK(1);
            {   // phase A, lane = step (IntegrateMotion's parameter-dependent part, geometry.cc:34-45)
                const Quat q = {f.qw, f.qx, f.qy, f.qz};
                const double lc[3] = {f.ax + bl[0], f.ay + bl[1], f.az + bl[2]};
                double rot[3];
                quat_rotate(q, lc, rot);
                for (int k = 0; k < 3; k++) G.sc[lane][k] = (rot[k] + bg[k]) * f.dt;
                G.sc[lane][3] = f.dt;
            }
            __syncthreads();
            CP_TICK(2);
            // phase S: v += a*dt ; travel += dt*v  (velocity.cc:99-108)
            fwd_chain(G.sc, comp, n, vk, tk);
            __syncthreads();
            CP_TICK(3);
        }
        const double travel[3] = {__shfl(tk, 0), __shfl(tk, 1), __shfl(tk, 2)};
        const double tn = sqrt(dot3(travel, travel));
        const double diff = tn - G.refDist[r];
        result += diff * diff;
        double d[3];
        for (int k = 0; k < 3; k++) d[k] = ((2.0 * diff) * travel[k]) / (tn + 1e-5);
        for (int c0 = 0; c0 < cnt; c0 += CB_CHUNK) {
            const int n = min(CB_CHUNK, cnt - c0);
            CP_TICK(4);
            const BwdRegs b = bn;
            bn = load_bwd(G.B, G.S, bnext + lane);
            bnext = G.chunks[++bi + 1];
            CP_TICK(5);
            for (int k = 0; k < 3; k++) {             // phase A, lane = step (velocity.cc:133-163)
                G.sp[lane][k] = b.c1 * d[k];
                const double row[3] = {b.m[3 * k], b.m[3 * k + 1], b.m[3 * k + 2]};
                G.sp[lane][3 + k] = dot3(row, d);
                G.sp[lane][6 + k] = b.dt * d[k];
            }
            __syncthreads();
            CP_TICK(6);
            bwd_chain(G.sp, acc, n, gj);
            __syncthreads();
            CP_TICK(7);
        }
    }
    CP_FLUSH;
    for (int k = 0; k < 9; k++) grad[k] = __shfl(gj, k) / G.totalSec;
    return result / G.totalSec;
}

// LBFGS.h:78-181 with Backtracking/Armijo (LineSearch.h:41-109); n = 9, m = 6, ftol 1e-4, 20 trials.
// Wave-uniform: every lane carries the same solver state.
__device__ int cal_lbfgs(const WinView& G, int lane, double* x, double* fx_out, double epsilon, int max_iterations)
{
    double s[CB_M][9], y[CB_M][9], ysh[CB_M], alpha[CB_M], xp[9], grad[9], gradp[9], drt[9];
    double fx = cal_eval(G, lane, x, grad);
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
            fx = cal_eval(G, lane, x, grad);
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
__global__ __launch_bounds__(64) void k_calibrate_windows(const WinDesc* __restrict__ wins, const double* __restrict__ dbl,
                                                        const int32_t* __restrict__ i32, int mode, const double* __restrict__ xin,
                                                        int max_iterations, double* __restrict__ xout, double* __restrict__ fxout,
                                                        double* __restrict__ gradout, int32_t* __restrict__ niter)
{
    __shared__ double sc[CB_CHUNK + 2 * CB_BATCH][4];      // + rows the one-batch-ahead reads may touch
    __shared__ double sp[CB_CHUNK + 2 * CB_BATCH][9];
    const WinDesc D = wins[blockIdx.x];
    const int lane = threadIdx.x;
    const size_t w = blockIdx.x;
    WinView G;
    G.F = dbl + D.fwd; G.B = dbl + D.bwd; G.refDist = dbl + D.refDist; G.refOff = i32 + D.refOff; G.chunks = i32 + D.chunks;
    G.S = D.S; G.nRef = D.nRef; G.totalSec = D.totalSec; G.sc = sc; G.sp = sp;
    double x[9], g[9], fx;
    if (mode == 1) {
        for (int i = 0; i < 9; i++) x[i] = xin[9 * w + i];
        fx = cal_eval(G, lane, x, g);
        if (lane < 9) gradout[9 * w + lane] = g[lane];
        if (lane == 0) fxout[w] = fx;
        return;
    }
    for (int i = 0; i < 9; i++) x[i] = 0.0;
    const int it = cal_lbfgs(G, lane, x, &fx, 1e-5, max_iterations);
    if (lane < 9) xout[9 * w + lane] = x[lane];
    if (lane == 0) { fxout[w] = fx; niter[w] = it; }
}

// ---- host: pack, launch ----

struct Packed {
    std::vector<WinDesc> wins;
    std::vector<double> dbl;
    std::vector<int32_t> i32;
};

// Streams of one window, in the order of velocity.cc:62-168 (both loops of a reference interval walk the
// same steps; the quantities below depend on the data only).
void window_streams(const Window& W, const Imu& M, const double* ref_v, std::vector<double>& fwd, std::vector<double>& bwd,
                    std::vector<double>& refDist, double* totalSec)
{
    Quat q = {1.0, 0.0, 0.0, 0.0};
    double twr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int64_t total_usec = 0;
    fwd.resize(W.steps.size() * CB_FWD); bwd.resize(W.steps.size() * CB_BWD);
    refDist.assign(W.refCnt.size(), 0.0);
    for (size_t i = 0; i < W.steps.size(); i++) {
        const Step& S = W.steps[i];
        const double* rr = M.rot + 3 * (size_t)M.evRot[S.event];
        const double* aa = M.acc + 3 * (size_t)M.evAcc[S.event];
        const double dt = (double)S.usec * 1e-6;
        double* f = &fwd[i * CB_FWD];
        f[0] = dt; f[1] = q.w; f[2] = q.x; f[3] = q.y; f[4] = q.z; f[5] = aa[0]; f[6] = aa[1]; f[7] = aa[2];
        refDist[W.refIdx[i]] += dt * ref_v[W.refIdx[i]];
        q = quat_mul(q, rotation_motion_to_quaternion(rr[0], rr[1], rr[2], dt));
        total_usec += S.usec;
        const double total_sec = (double)total_usec * 1e-6;
        double R[9];
        quat_matrix(q, R);
        for (int k = 0; k < 9; k++) twr[k] += R[k] * dt;
        double* b = &bwd[i * CB_BWD];
        b[0] = total_sec * dt; b[1] = dt;
        for (int k = 0; k < 3; k++) { b[2 + 3 * k] = dt * twr[k]; b[3 + 3 * k] = dt * twr[3 + k]; b[4 + 3 * k] = dt * twr[6 + k]; }
    }
    *totalSec = (double)total_usec * 1e-6;
}


// One window's streams as structure-of-arrays (a chunk of 64 consecutive steps of one field = 512 contiguous bytes).
void pack_window(const Window& W, const Imu& M, const double* ref_v, Packed& P)
{
    std::vector<double> fwd, bwd, rd;
    WinDesc D;
    window_streams(W, M, ref_v, fwd, bwd, rd, &D.totalSec);
    const size_t n = W.steps.size();
    D.S = (int32_t)((n + CB_CHUNK - 1) / CB_CHUNK * CB_CHUNK + CB_CHUNK);   // a chunk load may start at the last step
    D.nRef = (int32_t)W.refCnt.size();
    D.fwd = (int64_t)P.dbl.size();     P.dbl.resize(P.dbl.size() + (size_t)D.S * CB_FWD, 0.0);
    D.bwd = (int64_t)P.dbl.size();     P.dbl.resize(P.dbl.size() + (size_t)D.S * CB_BWD, 0.0);
    D.refDist = (int64_t)P.dbl.size(); P.dbl.insert(P.dbl.end(), rd.begin(), rd.end());
    D.refOff = (int64_t)P.i32.size();
    int32_t off = 0;
    for (int32_t c : W.refCnt) { P.i32.push_back(off); off += c; }
    P.i32.push_back(off);
    D.chunks = (int64_t)P.i32.size();
    off = 0;
    int32_t last = 0;
    for (int32_t c : W.refCnt) {
        for (int32_t c0 = 0; c0 < c; c0 += CB_CHUNK) { last = off + c0; P.i32.push_back(last); }
        off += c;
    }
    P.i32.push_back(last); P.i32.push_back(last); P.i32.push_back(last);     // the prefetcher reads up to two chunks past the end
    for (size_t i = 0; i < n; i++) {
        for (int f = 0; f < CB_FWD; f++) P.dbl[D.fwd + (size_t)f * D.S + i] = fwd[i * CB_FWD + f];
        for (int f = 0; f < CB_BWD; f++) P.dbl[D.bwd + (size_t)f * D.S + i] = bwd[i * CB_BWD + f];
    }
    P.wins.push_back(D);
}

void pack_windows(const std::vector<Window>& wins, const Imu& M, const std::vector<const double*>& ref_v, Packed& P)
{
    for (size_t w = 0; w < wins.size(); w++) pack_window(wins[w], M, ref_v[w], P);
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
    if (!dW.put(P.wins.data(), nw * sizeof(WinDesc)) || !dD.put(P.dbl.data(), P.dbl.size() * 8) || !dI.put(P.i32.data(), P.i32.size() * 4) ||
        !dXin.put(xin, xin ? nw * 72 : 0) || !dX.make(nw * 72) || !dF.make(nw * 8) || !dGr.make(nw * 72) || !dNi.make(nw * 4))
        return pg_ctx_fail(c, PGORB_E_HIP, "device allocation / upload for the calibration windows failed");
    const double t1 = now_s();
    hipLaunchKernelGGL(k_calibrate_windows, dim3((unsigned)nw), dim3(64), 0, 0, (const WinDesc*)dW.p, (const double*)dD.p, (const int32_t*)dI.p,
                       mode, (const double*)dXin.p, max_iters, (double*)dX.p, (double*)dF.p, (double*)dGr.p, (int32_t*)dNi.p);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_calibrate_windows failed");
#ifdef PGORB_CALIB_PROF
    {
        unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(cb_prof), sizeof(h));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(cb_prof), z, sizeof(z));
        fprintf(stderr, "[calib prof] cycles: rest/interval-end %llu | F: cursor+issue %llu, phase A %llu, phase S %llu | B: wait+d %llu, cursor+issue %llu, phase A %llu, phase S %llu\n",
                h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
#endif
    if (timing_on())
        fprintf(stderr, "[calib] %zu windows, %.1f MB of streams: upload %.3f s, solver kernel %.1f us\n", nw, P.dbl.size() * 8e-6, t1 - t0, (now_s() - t1) * 1e6);
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
    pack_window(wins[0], M, gps_velocity, P);
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
    // fit_motion.cc:196-246 with the fitted parameters: IntegrateTrajectory (velocity.cc:200-253) per window
    Imu M;
    merge_imu(rotations, rot_time_usec, n_rot, accelerations, acc_time_usec, n_acc, M);
    const int nev = (int)M.time.size();
    std::vector<std::vector<double>> lists(nev);                           // integrated_velocities (std::map keyed by event)
    double ksum[3] = {0, 0, 0}, krem[3] = {0, 0, 0};
    std::vector<Quat> ori(nev); std::vector<double> vel((size_t)nev * 3); std::vector<uint8_t> has(nev);
    for (int w = 0, start = 0; start < n_gps; start += locations_shift_step, w++) {
        const int end = std::min(start + locations_batch_size, n_gps);
        Window W;
        make_window(gps_time_usec + start, end - start, M, W);
        const double* p = &x[(size_t)w * 9];
        Quat q = {1.0, 0.0, 0.0, 0.0};
        double v[3] = {p[6], p[7], p[8]};
        std::fill(has.begin(), has.end(), 0);
        for (const Step& S : W.steps) {
            const double* rr = rotations + 3 * (size_t)M.evRot[S.event];
            const double* aa = accelerations + 3 * (size_t)M.evAcc[S.event];
            const double dt = (double)S.usec * 1e-6;
            const double lc[3] = {aa[0] + p[3], aa[1] + p[4], aa[2] + p[5]};
            double rot[3];
            quat_rotate(q, lc, rot);
            for (int k = 0; k < 3; k++) v[k] = v[k] + (rot[k] + p[k]) * dt;
            q = quat_mul(q, rotation_motion_to_quaternion(rr[0], rr[1], rr[2], dt));
            ori[S.event] = q; has[S.event] = 1;
            for (int k = 0; k < 3; k++) vel[3 * (size_t)S.event + k] = v[k];
        }
        double min_rotation_cos = 1.0;
        for (int e = 0; e < nev; e++) if (has[e]) {
            lists[e].push_back(sqrt(dot3(&vel[3 * (size_t)e], &vel[3 * (size_t)e])));
            min_rotation_cos = std::min(min_rotation_cos, std::abs(ori[e].w));
        }
        if (acos(min_rotation_cos) >= forward_axis_inference_min_rotation_rad)
            for (int e = 0; e < nev; e++) if (has[e] && sqrt(dot3(&vel[3 * (size_t)e], &vel[3 * (size_t)e])) >= forward_axis_inference_min_velocity_m_s) {
                const Quat inv = {ori[e].w, -ori[e].x, -ori[e].y, -ori[e].z};
                double vl[3];
                quat_rotate(inv, &vel[3 * (size_t)e], vl);
                for (int k = 0; k < 3; k++) {                              // KahanSum::add (math.hpp:13-19)
                    const double proposed = vl[k] + krem[k], updated = ksum[k] + proposed, actual = updated - ksum[k];
                    krem[k] = proposed - actual; ksum[k] = updated;
                }
            }
    }
    std::vector<double> avg, tsec;
    int n = 0;
    for (int e = 0; e < nev; e++) if (!lists[e].empty()) {
        out_time_usec[n] = M.time[e];
        tsec.push_back((double)(out_time_usec[n] - out_time_usec[0]) * 1e-6);
        double sum = 0.0;
        for (double vv : lists[e]) sum += vv;
        avg.push_back(sum / lists[e].size());
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

// calib.hip -- K9: fit_motion's sliding-window accelerometer calibration, every window's L-BFGS
// fit running at once on the GPU (BASELINE.json configs[4]).
//
// Reference: for every window of 40 GPS fixes (step 5) fit_motion builds an AccelerometerCalibrator
// and minimises its 9-parameter loss with LBFGSpp (src/fit_motion.cc:151-246,
// src/calibration/velocity.cc:42-180, thirdparty/LBFGS/LBFGS.h:78-181, LBFGS/LineSearch.h:41-109);
// one loss evaluation integrates every IMU sample of the window in time order
// (src/geometry/geometry.cc:24-53), ~10^4 dependent steps, and the fits run one after another.
//
// Here: windows are independent, so ONE LANE PER WINDOW runs the whole solver, 64 windows per wave,
// all waves concurrently.  The arithmetic inside a window stays in the reference's order (the sums
// are sequential in time; reordering them changes the doubles and the JSON is diffed bit for bit),
// so the parallelism is across windows only.  What does not depend on the nine parameters is
// hoisted out of the solver and computed once per window on the host, with the same operations in
// the same order as the reference's loop body:
//   forward stream  (8 doubles / step): dt, orientation BEFORE the step (w,x,y,z), raw acceleration
//   backward stream (11 doubles / step): total_time_sec*dt, dt, dt * total_time_weighted_rotation^T (3x3)
//   per GPS interval: reference_distance, number of steps
// (RotationMotionToQuaternion's sin/cos therefore run in the host libm, as in the reference.)
// Streams are interleaved by lane -- element (step, field) of the 64 windows of a group are 64
// consecutive doubles -- so every load of the solver is one fully coalesced 512-byte request; all
// lanes of a wave walk the same (interval, step) index, lanes whose interval is shorter idle.
// L-BFGS history (6 x 2 x 9 doubles per lane) lives in scratch (private memory is lane-interleaved
// by the hardware, i.e. also coalesced).  fp64 add/mul/div/sqrt are IEEE on both sides, contraction
// is off (Makefile), so a lane reproduces the CPU run of its window bit for bit.
//
// Eigen reduction orders (the parity contract shared with oracle/calib_oracle.c, E0-E5 there):
// 3-vectors t0 + (t1 + t2); the solver's 9-vectors in SSE2 packet order.
#include "pgorb_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

int pg_ctx_fail(pgorb_ctx* c, int code, const char* msg);
int pg_ctx_device(pgorb_ctx* c);

namespace {

#define CB_LANES 64
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

struct GroupView {
    const double* fwd;        // [step][CB_FWD][64]
    const double* bwd;        // [step][CB_BWD][64]
    const double* refDist;    // [r][64]
    const int32_t* refCnt;    // [r][64]
    const int32_t* refOff;    // [r]   first step slot of interval r (the same for all lanes)
    const int32_t* nRef;      // [64]
    const double* totalSec;   // [64]
};

__device__ inline double dot9(const double* a, const double* b)       // Eigen SSE2 redux order (contract E2)
{
    double l0 = a[0] * b[0], l1 = a[1] * b[1], m0 = a[2] * b[2], m1 = a[3] * b[3];
    l0 = l0 + a[4] * b[4]; l1 = l1 + a[5] * b[5];
    m0 = m0 + a[6] * b[6]; m1 = m1 + a[7] * b[7];
    l0 = l0 + m0; l1 = l1 + m1;
    return (l0 + l1) + a[8] * b[8];
}

__device__ double cal_eval(const GroupView& G, int lane, const double* x, double* grad)
{
    const double bg[3] = {x[0], x[1], x[2]}, bl[3] = {x[3], x[4], x[5]};
    double v[3] = {x[6], x[7], x[8]};
    double g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double result = 0;
    const int nref = G.nRef[lane];
    for (int r = 0; r < nref; r++) {
        const int cnt = G.refCnt[r * CB_LANES + lane];
        const size_t off = (size_t)G.refOff[r];
        double travel[3] = {0, 0, 0};
        const double* p = G.fwd + off * (CB_FWD * CB_LANES) + lane;
        for (int i = 0; i < cnt; i++, p += CB_FWD * CB_LANES) {
            const double dt = p[0];
            const Quat q = {p[CB_LANES], p[2 * CB_LANES], p[3 * CB_LANES], p[4 * CB_LANES]};
            const double lc[3] = {p[5 * CB_LANES] + bl[0], p[6 * CB_LANES] + bl[1], p[7 * CB_LANES] + bl[2]};
            double rot[3];
            quat_rotate(q, lc, rot);
            for (int k = 0; k < 3; k++) {
                const double glob = rot[k] + bg[k];
                v[k] = v[k] + glob * dt;
                travel[k] += dt * v[k];
            }
        }
        const double tn = sqrt(dot3(travel, travel));
        const double diff = tn - G.refDist[r * CB_LANES + lane];
        result += diff * diff;
        double d[3];
        for (int k = 0; k < 3; k++) d[k] = ((2.0 * diff) * travel[k]) / (tn + 1e-5);
        const double* b = G.bwd + off * (CB_BWD * CB_LANES) + lane;
        for (int i = 0; i < cnt; i++, b += CB_BWD * CB_LANES) {
            const double c1 = b[0], dt = b[CB_LANES];
            for (int k = 0; k < 3; k++) {
                g[k] += c1 * d[k];
                const double row[3] = {b[(2 + 3 * k) * CB_LANES], b[(3 + 3 * k) * CB_LANES], b[(4 + 3 * k) * CB_LANES]};
                g[3 + k] += dot3(row, d);
                g[6 + k] += dt * d[k];
            }
        }
    }
    const double total = G.totalSec[lane];
    for (int k = 0; k < 9; k++) grad[k] = g[k] / total;
    return result / total;
}

// LBFGS.h:78-181 with Backtracking/Armijo (LineSearch.h:41-109); n = 9, m = 6, ftol 1e-4, 20 trials.
__device__ int cal_lbfgs(const GroupView& G, int lane, double* x, double* fx_out, double epsilon, int max_iterations)
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

struct GroupDesc {            // one per wave: offsets (in elements) into the flat device arrays
    int64_t fwd, bwd, refDist, refCnt, refOff;
    int32_t lanes, pad;
};

// mode 0: L-BFGS from x = 0 (fit_motion.cc:186-190).  mode 1: one evaluation at xin (tests).
__global__ __launch_bounds__(CB_LANES) void k_calibrate_windows(const GroupDesc* __restrict__ groups, const double* __restrict__ dbl,
                                                              const int32_t* __restrict__ i32, const int32_t* __restrict__ nRef,
                                                              const double* __restrict__ totalSec, int mode, const double* __restrict__ xin,
                                                              int max_iterations, double* __restrict__ xout, double* __restrict__ fxout,
                                                              double* __restrict__ gradout, int32_t* __restrict__ niter)
{
    const GroupDesc D = groups[blockIdx.x];
    const int lane = threadIdx.x;
    if (lane >= D.lanes) return;
    const size_t w = (size_t)blockIdx.x * CB_LANES + lane;
    GroupView G;
    G.fwd = dbl + D.fwd; G.bwd = dbl + D.bwd; G.refDist = dbl + D.refDist;
    G.refCnt = i32 + D.refCnt; G.refOff = i32 + D.refOff;
    G.nRef = nRef + (size_t)blockIdx.x * CB_LANES; G.totalSec = totalSec + (size_t)blockIdx.x * CB_LANES;
    double x[9], fx;
    if (mode == 1) {
        double grad[9];
        for (int i = 0; i < 9; i++) x[i] = xin[9 * w + i];
        fx = cal_eval(G, lane, x, grad);
        for (int i = 0; i < 9; i++) gradout[9 * w + i] = grad[i];
        fxout[w] = fx;
        return;
    }
    for (int i = 0; i < 9; i++) x[i] = 0.0;
    const int it = cal_lbfgs(G, lane, x, &fx, 1e-5, max_iterations);
    for (int i = 0; i < 9; i++) xout[9 * w + i] = x[i];
    fxout[w] = fx;
    niter[w] = it;
}

// ---- host: pack, launch ----

struct Packed {
    std::vector<GroupDesc> groups;
    std::vector<double> dbl;
    std::vector<int32_t> i32, nRef;
    std::vector<double> totalSec;
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

void pack_windows(const std::vector<Window>& wins, const Imu& M, const std::vector<const double*>& ref_v, Packed& P)
{
    const int nw = (int)wins.size(), ng = (nw + CB_LANES - 1) / CB_LANES;
    P.groups.resize(ng);
    P.nRef.assign((size_t)ng * CB_LANES, 0);
    P.totalSec.assign((size_t)ng * CB_LANES, 1.0);
    std::vector<double> fwd, bwd, rd;
    for (int g = 0; g < ng; g++) {
        const int w0 = g * CB_LANES, lanes = std::min(CB_LANES, nw - w0);
        int maxRef = 0;
        for (int l = 0; l < lanes; l++) maxRef = std::max(maxRef, (int)wins[w0 + l].refCnt.size());
        std::vector<int32_t> refOff(maxRef + 1, 0);
        for (int r = 0; r < maxRef; r++) {
            int mx = 0;
            for (int l = 0; l < lanes; l++) if (r < (int)wins[w0 + l].refCnt.size()) mx = std::max(mx, wins[w0 + l].refCnt[r]);
            refOff[r + 1] = refOff[r] + mx;
        }
        const size_t slots = (size_t)refOff[maxRef];
        GroupDesc& D = P.groups[g];
        D.lanes = lanes; D.pad = 0;
        D.fwd = (int64_t)P.dbl.size();     P.dbl.resize(P.dbl.size() + slots * CB_FWD * CB_LANES, 0.0);
        D.bwd = (int64_t)P.dbl.size();     P.dbl.resize(P.dbl.size() + slots * CB_BWD * CB_LANES, 0.0);
        D.refDist = (int64_t)P.dbl.size(); P.dbl.resize(P.dbl.size() + (size_t)maxRef * CB_LANES, 0.0);
        D.refCnt = (int64_t)P.i32.size();  P.i32.resize(P.i32.size() + (size_t)maxRef * CB_LANES, 0);
        D.refOff = (int64_t)P.i32.size();  P.i32.insert(P.i32.end(), refOff.begin(), refOff.end());
        for (int l = 0; l < lanes; l++) {
            const Window& W = wins[w0 + l];
            double total;
            window_streams(W, M, ref_v[w0 + l], fwd, bwd, rd, &total);
            P.nRef[(size_t)g * CB_LANES + l] = (int32_t)W.refCnt.size();
            P.totalSec[(size_t)g * CB_LANES + l] = total;
            size_t si = 0;
            for (size_t r = 0; r < W.refCnt.size(); r++) {
                P.i32[D.refCnt + r * CB_LANES + l] = W.refCnt[r];
                P.dbl[D.refDist + r * CB_LANES + l] = rd[r];
                for (int i = 0; i < W.refCnt[r]; i++, si++) {
                    const size_t slot = (size_t)refOff[r] + i;
                    for (int f = 0; f < CB_FWD; f++) P.dbl[D.fwd + (slot * CB_FWD + f) * CB_LANES + l] = fwd[si * CB_FWD + f];
                    for (int f = 0; f < CB_BWD; f++) P.dbl[D.bwd + (slot * CB_BWD + f) * CB_LANES + l] = bwd[si * CB_BWD + f];
                }
            }
        }
    }
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool put(const void* src, size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess && (!bytes || hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) == hipSuccess); }
    bool make(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess; }
};

int run_windows(pgorb_ctx* c, const Packed& P, int nw, int mode, const double* xin, int max_iters, double* x, double* fx, double* grad, int32_t* niter)
{
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    const size_t slots = P.groups.size() * CB_LANES;
    DevBuf dG, dD, dI, dN, dT, dXin, dX, dF, dGr, dNi;
    std::vector<double> xpad(slots * 9, 0.0);
    if (xin) memcpy(xpad.data(), xin, sizeof(double) * 9 * (size_t)nw);
    if (!dG.put(P.groups.data(), P.groups.size() * sizeof(GroupDesc)) || !dD.put(P.dbl.data(), P.dbl.size() * 8) ||
        !dI.put(P.i32.data(), P.i32.size() * 4) || !dN.put(P.nRef.data(), P.nRef.size() * 4) || !dT.put(P.totalSec.data(), P.totalSec.size() * 8) ||
        !dXin.put(xpad.data(), xpad.size() * 8) || !dX.make(slots * 72) || !dF.make(slots * 8) || !dGr.make(slots * 72) || !dNi.make(slots * 4))
        return pg_ctx_fail(c, PGORB_E_HIP, "device allocation / upload for the calibration windows failed");
    hipLaunchKernelGGL(k_calibrate_windows, dim3((unsigned)P.groups.size()), dim3(CB_LANES), 0, 0, (const GroupDesc*)dG.p, (const double*)dD.p,
                       (const int32_t*)dI.p, (const int32_t*)dN.p, (const double*)dT.p, mode, (const double*)dXin.p, max_iters,
                       (double*)dX.p, (double*)dF.p, (double*)dGr.p, (int32_t*)dNi.p);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_calibrate_windows failed");
    bool ok = hipMemcpy(fx, dF.p, sizeof(double) * (size_t)nw, hipMemcpyDeviceToHost) == hipSuccess;
    if (mode == 1) ok = ok && hipMemcpy(grad, dGr.p, sizeof(double) * 9 * (size_t)nw, hipMemcpyDeviceToHost) == hipSuccess;
    else ok = ok && hipMemcpy(x, dX.p, sizeof(double) * 9 * (size_t)nw, hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(niter, dNi.p, sizeof(int32_t) * (size_t)nw, hipMemcpyDeviceToHost) == hipSuccess;
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
    pack_windows(wins, M, refv, P);
    return run_windows(c, P, (int)wins.size(), 0, nullptr, optimization_iters, x, residual, nullptr, niter);
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
    wins.resize(n_points, wins[0]);                          // the same calibrator evaluated at n_points parameter vectors
    std::vector<const double*> refv(n_points, gps_velocity);
    Packed P;
    pack_windows(wins, M, refv, P);
    return run_windows(c, P, n_points, 1, xin, 0, nullptr, fx, grad, nullptr);
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

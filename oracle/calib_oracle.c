/* calib_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, never linked into the product) for the
 * velocity half of pilotguru's fit_motion: sliding-window accelerometer calibration with L-BFGS
 * (BASELINE.json configs[4]).
 *
 * Restates in plain C, structure for structure:
 *   src/interpolation/align_time_series.cc:29-118   MergeTimeSeries (two components), GetEffectiveTimeStamp
 *   src/interpolation/align_time_series.cc:150-195  MakeInterpolationIntervals
 *   src/geometry/geometry.cc:6-53                   RotationMotionToQuaternion, IntegrateMotion
 *   src/calibration/velocity.cc:42-180              AccelerometerCalibrator::eval
 *   src/calibration/velocity.cc:200-253             AccelerometerCalibrator::IntegrateTrajectory
 *   thirdparty/LBFGS/LBFGS.h:78-181, LBFGS/LineSearch.h:41-109   LBFGSSolver::minimize, Backtracking (Armijo)
 *   src/fit_motion.cc:151-290                       ComputeAndSaveForwardVelocitiesFromImu
 *   include/math/math.hpp:8-25                      KahanSum
 *
 * Eigen (libeigen3-dev of ubuntu 16.04, un-vendored) supplies the vector arithmetic.  The reference
 * is built with -O3 -march=native (CMakeLists.txt:19-20), so its own double results depend on the
 * build machine (FMA contraction, AVX packets).  PARITY CONTRACT used here and in the product
 * (PARITY UNPINNED -- no Eigen in this image, no golden vectors in the reference):
 *   E0  no FMA contraction anywhere;
 *   E1  fixed-size 3-vector reductions (dot, squaredNorm, rows of a 3x3 product) associate as
 *       t0 + (t1 + t2)  (Eigen's unrolled non-vectorised redux splits the range in halves);
 *   E2  dynamic 9-vector reductions (the solver's dot / norm) follow Eigen's SSE2 linear
 *       vectorised redux: lanes p0 = (t0,t1) += (t4,t5), p1 = (t2,t3) += (t6,t7), p0 += p1,
 *       lane0 + lane1, then + t8;
 *   E3  quaternion product = the generic formula, terms left to right;
 *   E4  Quaternion::_transformVector: uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv;
 *   E5  Quaternion::toRotationMatrix as in Eigen/src/Geometry/Quaternion.h (tx = 2x ... ). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double w, x, y, z; } quat;
typedef struct { double v[3]; } vec3;

static double dot3(const double* a, const double* b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }   /* E1 */
static double norm3(const double* a) { return sqrt(dot3(a, a)); }

static double dot9(const double* a, const double* b)                                                           /* E2 */
{
    double l0 = a[0] * b[0], l1 = a[1] * b[1], m0 = a[2] * b[2], m1 = a[3] * b[3];
    l0 = l0 + a[4] * b[4]; l1 = l1 + a[5] * b[5];
    m0 = m0 + a[6] * b[6]; m1 = m1 + a[7] * b[7];
    l0 = l0 + m0; l1 = l1 + m1;
    return (l0 + l1) + a[8] * b[8];
}
static double norm9(const double* a) { return sqrt(dot9(a, a)); }

static quat qmul(quat a, quat b)                                                                               /* E3 */
{
    quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

static void qrot(quat q, const double* v, double* out)                                                         /* E4 */
{
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    double c[3];
    int i;
    for (i = 0; i < 3; i++) uv[i] += uv[i];
    c[0] = q.y * uv[2] - q.z * uv[1]; c[1] = q.z * uv[0] - q.x * uv[2]; c[2] = q.x * uv[1] - q.y * uv[0];
    for (i = 0; i < 3; i++) out[i] = (v[i] + q.w * uv[i]) + c[i];
}

static void qmat(quat q, double* R)                                                                            /* E5, row-major */
{
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

/* geometry.cc:6-22 */
static quat rotation_motion_to_quaternion(double rx, double ry, double rz, double duration_sec)
{
    const double rate = sqrt(rx * rx + ry * ry + rz * rz);
    const double half_theta = rate * duration_sec * 0.5;
    const double s = sin(half_theta) / (rate + 1e-30);
    quat q;
    q.w = cos(half_theta); q.x = rx * s; q.y = ry * s; q.z = rz * s;
    return q;
}

/* ---- align_time_series.cc ---- */

typedef struct { int64_t ref_end, interp_end, start_usec, end_usec; } interval_t;
typedef struct { int n; interval_t* iv; } interval_list;

typedef struct {
    /* inputs (borrowed) */
    const double* ref_v; const int64_t* ref_t; int n_ref;
    const double* rot; const int64_t* rot_t; int n_rot;
    const double* acc; const int64_t* acc_t; int n_acc;
    /* merged series: indices into rot / acc per event */
    int n_events; int32_t* ev_rot; int32_t* ev_acc;
    /* per reference point, its intervals */
    interval_list* per_ref;
} calibrator;

static size_t lower_bound64(const int64_t* a, size_t n, int64_t v)
{
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

/* MergeTimeSeries({rot_t, acc_t}), :29-118.  Returns 0, or -1 where the reference CHECK-fails. */
static int merge_time_series(calibrator* c)
{
    const int64_t* T[2] = {c->rot_t, c->acc_t};
    const size_t N[2] = {(size_t)c->n_rot, (size_t)c->n_acc};
    size_t cur[2], cap, k;
    int64_t start_time, end_time;
    c->n_events = 0; c->ev_rot = c->ev_acc = NULL;
    for (k = 0; k < 2; k++) {
        size_t i;
        if (N[k] == 0) return -1;
        for (i = 0; i + 1 < N[k]; i++) if (!(T[k][i] < T[k][i + 1])) return -1;
    }
    start_time = T[0][0] > T[1][0] ? T[0][0] : T[1][0];
    end_time = T[0][N[0] - 1] < T[1][N[1] - 1] ? T[0][N[0] - 1] : T[1][N[1] - 1];
    if (end_time < start_time) return 0;
    for (k = 0; k < 2; k++) {
        size_t idx = lower_bound64(T[k], N[k], start_time);
        if (idx >= N[k]) return -1;
        if (T[k][idx] > start_time) { if (idx == 0) return -1; cur[k] = idx - 1; }
        else cur[k] = idx;
    }
    cap = N[0] + N[1] + 1;
    c->ev_rot = (int32_t*)malloc(sizeof(int32_t) * cap);
    c->ev_acc = (int32_t*)malloc(sizeof(int32_t) * cap);
    for (;;) {
        int64_t next_times[2], next_time;
        c->ev_rot[c->n_events] = (int32_t)cur[0]; c->ev_acc[c->n_events] = (int32_t)cur[1]; c->n_events++;
        for (k = 0; k < 2; k++) {
            if (cur[k] + 1 >= N[k]) return 0;
            next_times[k] = T[k][cur[k] + 1];
        }
        next_time = next_times[0] < next_times[1] ? next_times[0] : next_times[1];
        for (k = 0; k < 2; k++) if (T[k][cur[k] + 1] == next_time) cur[k]++;
    }
}

static int64_t merged_event_time(const calibrator* c, int e)                /* GetEffectiveTimeStamp :120-133 */
{
    int64_t a = c->rot_t[c->ev_rot[e]], b = c->acc_t[c->ev_acc[e]];
    return a > b ? a : b;
}

/* MakeInterpolationIntervals(reference timestamps, merged event times), :150-195 */
static int make_intervals(calibrator* c)
{
    int64_t latest_ts;
    int reference_idx, interpolation_idx = 0, i;
    int64_t* imu = (int64_t*)malloc(sizeof(int64_t) * (size_t)(c->n_events > 0 ? c->n_events : 1));
    for (i = 0; i < c->n_events; i++) imu[i] = merged_event_time(c, i);
    for (i = 0; i + 1 < c->n_ref; i++) if (!(c->ref_t[i] < c->ref_t[i + 1])) { free(imu); return -1; }
    for (i = 0; i + 1 < c->n_events; i++) if (!(imu[i] < imu[i + 1])) { free(imu); return -1; }
    if (c->n_events == 0 || c->n_ref == 0) { free(imu); return -1; }     /* front() of an empty vector */
    c->per_ref = (interval_list*)calloc((size_t)c->n_ref, sizeof(interval_list));
    latest_ts = imu[0] < c->ref_t[0] ? imu[0] : c->ref_t[0];
    for (reference_idx = 0; reference_idx < c->n_ref; ++reference_idx) {
        const int64_t reference_ts = c->ref_t[reference_idx];
        interval_list* L = &c->per_ref[reference_idx];
        int cap = 16;
        L->iv = (interval_t*)malloc(sizeof(interval_t) * (size_t)cap);
        while (interpolation_idx < c->n_events && imu[interpolation_idx] <= reference_ts) {
            const int64_t interpolation_ts = imu[interpolation_idx];
            if (interpolation_ts > latest_ts && interpolation_idx > 0 && reference_idx > 0) {
                if (L->n == cap) { cap *= 2; L->iv = (interval_t*)realloc(L->iv, sizeof(interval_t) * (size_t)cap); }
                L->iv[L->n].ref_end = reference_idx; L->iv[L->n].interp_end = interpolation_idx;
                L->iv[L->n].start_usec = latest_ts; L->iv[L->n].end_usec = interpolation_ts; L->n++;
            }
            latest_ts = interpolation_ts;
            ++interpolation_idx;
        }
        if (interpolation_idx > 0 && reference_idx > 0 && interpolation_idx < c->n_events && reference_ts > latest_ts) {
            if (L->n == cap) { cap *= 2; L->iv = (interval_t*)realloc(L->iv, sizeof(interval_t) * (size_t)cap); }
            L->iv[L->n].ref_end = reference_idx; L->iv[L->n].interp_end = interpolation_idx;
            L->iv[L->n].start_usec = latest_ts; L->iv[L->n].end_usec = reference_ts; L->n++;
        }
        latest_ts = reference_ts;
    }
    free(imu);
    return 0;
}

static int calibrator_init(calibrator* c, const double* ref_v, const int64_t* ref_t, int n_ref,
                           const double* rot, const int64_t* rot_t, int n_rot,
                           const double* acc, const int64_t* acc_t, int n_acc)
{
    memset(c, 0, sizeof(*c));
    c->ref_v = ref_v; c->ref_t = ref_t; c->n_ref = n_ref;
    c->rot = rot; c->rot_t = rot_t; c->n_rot = n_rot;
    c->acc = acc; c->acc_t = acc_t; c->n_acc = n_acc;
    if (merge_time_series(c)) return -1;
    return make_intervals(c);
}

static void calibrator_free(calibrator* c)
{
    int i;
    if (c->per_ref) { for (i = 0; i < c->n_ref; i++) free(c->per_ref[i].iv); free(c->per_ref); }
    free(c->ev_rot); free(c->ev_acc);
    memset(c, 0, sizeof(*c));
}

/* ---- velocity.cc:42-180  AccelerometerCalibrator::eval ---- */

typedef struct { quat orientation; double velocity[3]; int64_t duration_usec; } outcome_t;

static outcome_t integrate_motion(quat start_orientation, const double* start_velocity, quat raw_rotation,
                                  const double* raw_acceleration, const double* global_bias, const double* local_bias,
                                  int64_t duration_usec)                                  /* geometry.cc:24-53 */
{
    const double duration_sec = (double)duration_usec * 1e-6;
    double local_cal[3], rotated[3], global[3];
    outcome_t o;
    int i;
    for (i = 0; i < 3; i++) local_cal[i] = raw_acceleration[i] + local_bias[i];
    qrot(start_orientation, local_cal, rotated);
    for (i = 0; i < 3; i++) global[i] = rotated[i] + global_bias[i];
    for (i = 0; i < 3; i++) o.velocity[i] = start_velocity[i] + global[i] * duration_sec;
    o.orientation = qmul(start_orientation, raw_rotation);
    o.duration_usec = duration_usec;
    return o;
}

static double calibrator_eval(const calibrator* c, const double* in, double* gradient)
{
    const double* global_bias = in; const double* local_bias = in + 3;
    double result = 0, integrated_velocity[3] = {in[6], in[7], in[8]};
    double twr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                          /* total_time_weighted_rotation */
    quat integrated_rotation = {1.0, 0.0, 0.0, 0.0};
    int64_t total_time_usec = 0;
    double total_time_sec;
    int r, i, k, maxn = 0;
    outcome_t* outcomes;
    for (i = 0; i < 9; i++) gradient[i] = 0.0;
    for (r = 0; r < c->n_ref; r++) if (c->per_ref[r].n > maxn) maxn = c->per_ref[r].n;
    outcomes = (outcome_t*)malloc(sizeof(outcome_t) * (size_t)(maxn > 0 ? maxn : 1));
    for (r = 0; r < c->n_ref; r++) {
        const interval_list* L = &c->per_ref[r];
        double integrated_travel[3] = {0, 0, 0}, reference_distance = 0, distance_diff, d[3], tn;
        for (i = 0; i < L->n; i++) {
            const interval_t* iv = &L->iv[i];
            const double* rr = c->rot + 3 * (size_t)c->ev_rot[iv->interp_end];
            const double* aa = c->acc + 3 * (size_t)c->ev_acc[iv->interp_end];
            const double duration_sec = (double)(iv->end_usec - iv->start_usec) * 1e-6;
            const quat raw_rotation = rotation_motion_to_quaternion(rr[0], rr[1], rr[2], duration_sec);
            const outcome_t o = integrate_motion(integrated_rotation, integrated_velocity, raw_rotation, aa,
                                                 global_bias, local_bias, iv->end_usec - iv->start_usec);
            outcomes[i] = o;
            integrated_rotation = o.orientation;
            for (k = 0; k < 3; k++) integrated_velocity[k] = o.velocity[k];
            for (k = 0; k < 3; k++) integrated_travel[k] += duration_sec * o.velocity[k];
            reference_distance += duration_sec * c->ref_v[iv->ref_end];
        }
        tn = norm3(integrated_travel);
        distance_diff = tn - reference_distance;
        result += distance_diff * distance_diff;
        for (k = 0; k < 3; k++) d[k] = ((2.0 * distance_diff) * integrated_travel[k]) / (norm3(integrated_travel) + 1e-5);
        for (i = 0; i < L->n; i++) {
            const outcome_t* o = &outcomes[i];
            const double interval_sec = (double)o->duration_usec * 1e-6;
            double R[9], tmp[9], dl[3];
            total_time_usec += o->duration_usec;
            total_time_sec = (double)total_time_usec * 1e-6;
            for (k = 0; k < 3; k++) gradient[k] += total_time_sec * interval_sec * d[k];
            qmat(o->orientation, R);
            for (k = 0; k < 9; k++) twr[k] += R[k] * interval_sec;
            /* interval_sec * twr^T * d : the scaled transpose is evaluated, then a 3x3 * 3x1 product */
            for (k = 0; k < 3; k++) { tmp[3 * k] = interval_sec * twr[k]; tmp[3 * k + 1] = interval_sec * twr[3 + k]; tmp[3 * k + 2] = interval_sec * twr[6 + k]; }
            for (k = 0; k < 3; k++) dl[k] = dot3(tmp + 3 * k, d);
            for (k = 0; k < 3; k++) gradient[3 + k] += dl[k];
            for (k = 0; k < 3; k++) gradient[6 + k] += interval_sec * d[k];
        }
    }
    free(outcomes);
    total_time_sec = (double)total_time_usec * 1e-6;
    result /= total_time_sec;
    for (i = 0; i < 9; i++) gradient[i] /= total_time_sec;
    return result;
}

/* porc_calibrator_eval: one evaluation (tests: gradient vs finite differences, GPU eval parity) */
int porc_calibrator_eval(const double* ref_v, const int64_t* ref_t, int n_ref, const double* rot, const int64_t* rot_t, int n_rot,
                         const double* acc, const int64_t* acc_t, int n_acc, const double* x, double* fx, double* grad)
{
    calibrator c;
    if (calibrator_init(&c, ref_v, ref_t, n_ref, rot, rot_t, n_rot, acc, acc_t, n_acc)) { calibrator_free(&c); return -1; }
    *fx = calibrator_eval(&c, x, grad);
    calibrator_free(&c);
    return 0;
}

/* ---- LBFGS.h:78-181 + LineSearch.h:41-109, n = 9, default LBFGSParam except epsilon / max_iterations ---- */

#define LB_M 6
/* returns the iteration count; < 0: the line search threw (-2 step below min_step, -3 above max_step) */
static int lbfgs_minimize(const calibrator* c, double* x, double* fx_out, double epsilon, int max_iterations)
{
    const int n = 9, m = LB_M, max_linesearch = 20;
    const double ftol = 1e-4, min_step = 1e-20, max_step = 1e+20;
    double s[LB_M][9], y[LB_M][9], ys_hist[LB_M], alpha[LB_M], xp[9], grad[9], gradp[9], drt[9];
    double fx, xnorm, gnorm, step;
    int k = 1, end = 0, i, j, t;
    fx = calibrator_eval(c, x, grad);
    xnorm = norm9(x); gnorm = norm9(grad);
    if (gnorm <= epsilon * (xnorm > 1.0 ? xnorm : 1.0)) { *fx_out = fx; return 1; }
    for (i = 0; i < n; i++) drt[i] = -grad[i];
    step = 1.0 / norm9(drt);
    for (;;) {
        double ys, yy;
        int bound;
        memcpy(xp, x, sizeof(xp)); memcpy(gradp, grad, sizeof(gradp));
        {   /* Backtracking, Armijo */
            const double fx_init = fx, dg_init = dot9(grad, drt), dg_test = ftol * dg_init;
            int iter;
            for (iter = 0; iter < max_linesearch; iter++) {
                for (i = 0; i < n; i++) x[i] = xp[i] + step * drt[i];
                fx = calibrator_eval(c, x, grad);
                if (fx > fx_init + step * dg_test) { /* width = dec */ } else break;
                if (step < min_step) { *fx_out = fx; return -2; }
                if (step > max_step) { *fx_out = fx; return -3; }
                step *= 0.5;
            }
        }
        xnorm = norm9(x); gnorm = norm9(grad);
        if (gnorm <= epsilon * (xnorm > 1.0 ? xnorm : 1.0)) { *fx_out = fx; return k; }
        if (max_iterations != 0 && k >= max_iterations) { *fx_out = fx; return k; }
        for (i = 0; i < n; i++) { s[end][i] = x[i] - xp[i]; y[end][i] = grad[i] - gradp[i]; }
        ys = dot9(y[end], s[end]); yy = dot9(y[end], y[end]);
        ys_hist[end] = ys;
        for (i = 0; i < n; i++) drt[i] = -grad[i];
        bound = m < k ? m : k;
        end = (end + 1) % m;
        j = end;
        for (t = 0; t < bound; t++) {
            j = (j + m - 1) % m;
            alpha[j] = dot9(s[j], drt) / ys_hist[j];
            for (i = 0; i < n; i++) drt[i] -= alpha[j] * y[j][i];
        }
        { const double sc = ys / yy; for (i = 0; i < n; i++) drt[i] *= sc; }
        for (t = 0; t < bound; t++) {
            const double beta = dot9(y[j], drt) / ys_hist[j];
            const double ab = alpha[j] - beta;
            for (i = 0; i < n; i++) drt[i] += ab * s[j][i];
            j = (j + 1) % m;
        }
        step = 1.0;
        k++;
    }
}

/* ---- fit_motion.cc:151-246: the sliding-window fits ---- */

int porc_num_windows(int n_ref, int shift_step) { return n_ref <= 0 ? 0 : (n_ref + shift_step - 1) / shift_step; }

/* x_out[w][9], residual[w], niter[w]; returns the number of windows or < 0 */
int porc_fit_windows(const double* ref_v, const int64_t* ref_t, int n_ref, const double* rot, const int64_t* rot_t, int n_rot,
                     const double* acc, const int64_t* acc_t, int n_acc, int batch_size, int shift_step, int max_iters,
                     double* x_out, double* residual, int32_t* niter)
{
    int start, w = 0;
    for (start = 0; start < n_ref; start += shift_step, w++) {
        const int endi = start + batch_size < n_ref ? start + batch_size : n_ref;
        calibrator c;
        double x[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (calibrator_init(&c, ref_v + start, ref_t + start, endi - start, rot, rot_t, n_rot, acc, acc_t, n_acc)) { calibrator_free(&c); return -1; }
        niter[w] = lbfgs_minimize(&c, x, &residual[w], 1e-5, max_iters);
        memcpy(x_out + 9 * (size_t)w, x, sizeof(x));
        calibrator_free(&c);
    }
    return w;
}

/* ---- fit_motion.cc:151-290 complete: velocities per merged IMU event + forward axis ----
 * out_time_usec / out_velocity need n_rot + n_acc entries.  Returns the number of output points, < 0 on error. */
int porc_fit_motion_velocities(const double* ref_v, const int64_t* ref_t, int n_ref, const double* rot, const int64_t* rot_t, int n_rot,
                               const double* acc, const int64_t* acc_t, int n_acc, const double* vertical_axis,
                               int batch_size, int shift_step, int max_iters, double post_smoothing_sigma_sec,
                               double min_velocity, double min_rotation_rad,
                               int64_t* out_time_usec, double* out_velocity, double* forward_axis)
{
    extern int porc_smooth_time_series(const double*, const double*, int, const double*, int, double, double*);
    calibrator all;
    double ksum[3] = {0, 0, 0}, krem[3] = {0, 0, 0};
    double *vsum_list, *tsec, *avg;
    int32_t* cnt;
    int nev, start, i, k, npts = 0;
    /* per event: the velocities pushed by each window, in window order (std::map<size_t, vector<double>>) */
    int* lcap; double** lists;
    if (calibrator_init(&all, ref_v, ref_t, n_ref, rot, rot_t, n_rot, acc, acc_t, n_acc)) { calibrator_free(&all); return -1; }
    nev = all.n_events;
    cnt = (int32_t*)calloc((size_t)nev, sizeof(int32_t)); lcap = (int*)calloc((size_t)nev, sizeof(int)); lists = (double**)calloc((size_t)nev, sizeof(double*));
    for (start = 0; start < n_ref; start += shift_step) {
        const int endi = start + batch_size < n_ref ? start + batch_size : n_ref;
        calibrator c;
        double x[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, fx, min_rotation_cos = 1.0;
        outcome_t* res; uint8_t* has;
        quat integrated_rotation = {1.0, 0.0, 0.0, 0.0};
        double integrated_velocity[3];
        int r;
        if (calibrator_init(&c, ref_v + start, ref_t + start, endi - start, rot, rot_t, n_rot, acc, acc_t, n_acc)) { calibrator_free(&c); return -1; }
        if (lbfgs_minimize(&c, x, &fx, 1e-5, max_iters) < 0) { calibrator_free(&c); return -2; }
        /* IntegrateTrajectory(global, local, initial), velocity.cc:200-253 */
        res = (outcome_t*)malloc(sizeof(outcome_t) * (size_t)nev); has = (uint8_t*)calloc((size_t)nev, 1);
        for (k = 0; k < 3; k++) integrated_velocity[k] = x[6 + k];
        for (r = 0; r < c.n_ref; r++) for (i = 0; i < c.per_ref[r].n; i++) {
            const interval_t* iv = &c.per_ref[r].iv[i];
            const int idx = (int)iv->interp_end;
            const double* rr = rot + 3 * (size_t)c.ev_rot[idx];
            const double* aa = acc + 3 * (size_t)c.ev_acc[idx];
            const quat raw = rotation_motion_to_quaternion(rr[0], rr[1], rr[2], (double)(iv->end_usec - iv->start_usec) * 1e-6);
            const outcome_t o = integrate_motion(integrated_rotation, integrated_velocity, raw, aa, x, x + 3, iv->end_usec - iv->start_usec);
            integrated_rotation = o.orientation;
            for (k = 0; k < 3; k++) integrated_velocity[k] = o.velocity[k];
            if (!has[idx]) { res[idx] = o; has[idx] = 1; }
            else { res[idx].orientation = o.orientation; memcpy(res[idx].velocity, o.velocity, sizeof(o.velocity)); res[idx].duration_usec += o.duration_usec; }
        }
        for (i = 0; i < nev; i++) if (has[i]) {
            if (cnt[i] == lcap[i]) { lcap[i] = lcap[i] ? 2 * lcap[i] : 8; lists[i] = (double*)realloc(lists[i], sizeof(double) * (size_t)lcap[i]); }
            lists[i][cnt[i]++] = norm3(res[i].velocity);
            { const double aw = fabs(res[i].orientation.w); if (aw < min_rotation_cos) min_rotation_cos = aw; }
        }
        if (acos(min_rotation_cos) >= min_rotation_rad) {
            for (i = 0; i < nev; i++) if (has[i] && norm3(res[i].velocity) >= min_velocity) {
                quat inv = res[i].orientation; double vl[3];
                inv.x = -inv.x; inv.y = -inv.y; inv.z = -inv.z;
                qrot(inv, res[i].velocity, vl);
                for (k = 0; k < 3; k++) {                                  /* KahanSum<Vector3d>::add */
                    const double proposed = vl[k] + krem[k], updated = ksum[k] + proposed, actual = updated - ksum[k];
                    krem[k] = proposed - actual; ksum[k] = updated;
                }
            }
        }
        free(res); free(has);
        calibrator_free(&c);
    }
    vsum_list = (double*)malloc(sizeof(double) * (size_t)(nev > 0 ? nev : 1));
    tsec = (double*)malloc(sizeof(double) * (size_t)(nev > 0 ? nev : 1));
    avg = vsum_list;
    for (i = 0; i < nev; i++) if (cnt[i]) {
        double sum = 0.0;
        out_time_usec[npts] = merged_event_time(&all, i);
        tsec[npts] = (double)(out_time_usec[npts] - out_time_usec[0]) * 1e-6;
        for (k = 0; k < cnt[i]; k++) sum += lists[i][k];
        avg[npts] = sum / cnt[i];
        npts++;
    }
    if (npts && porc_smooth_time_series(avg, tsec, npts, tsec, npts, post_smoothing_sigma_sec, out_velocity)) npts = -3;
    {
        double f[3] = {ksum[0], ksum[1], ksum[2]}, dp, nn;
        dp = dot3(vertical_axis, f);
        for (k = 0; k < 3; k++) f[k] -= vertical_axis[k] * dp;
        nn = norm3(f) + 1e-5;
        for (k = 0; k < 3; k++) forward_axis[k] = f[k] / nn;
    }
    for (i = 0; i < nev; i++) free(lists[i]);
    free(lists); free(lcap); free(cnt); free(vsum_list); free(tsec);
    calibrator_free(&all);
    return npts;
}

/* include/math/math.hpp:8-25, n vectors of dim components (the form used inside porc_fit_motion_velocities) */
int porc_kahan_sum(const double* values, int n, int dim, double* sum)
{
    int i, k;
    for (k = 0; k < dim; k++) {
        double s = 0.0, rem = 0.0;
        for (i = 0; i < n; i++) {
            const double proposed = values[(size_t)i * dim + k] + rem, updated = s + proposed, actual = updated - s;
            rem = proposed - actual; s = updated;
        }
        sum[k] = s;
    }
    return 0;
}

/* post_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, never linked into the product) for the
 * trajectory post-processing of optical_trajectories: SURVEY.md §8 row f4.
 *
 * Restates, in plain C and matrix by matrix as the reference builds them:
 *   src/slam/smoothing.cc:11-47            SmoothHeadingDirections
 *   src/slam/smoothing.cc:49-97            NormalCdf, SmoothTimeSeries
 *   src/slam/track_image_sequence.cc:16-29 TrajectoryToPCA
 *   src/slam/horizontal_flatten.cc:7-63    ProjectDirections, ProjectTranslations,
 *                                          Projected2DDirectionsToTurnAngles
 * The OpenCV 2.4.9 routines underneath (getGaussianKernel, sepFilter2D, norm, PCA =
 * reduce + mulTransposed + eigen (Jacobi), gemm, Vec dot/cross) and Eigen's
 * Quaternion::_transformVector are third-party and absent from /root/reference and from
 * this image; their operation order is written here from the published 2.4 sources as
 * remembered.  PARITY UNPINNED for those: there is no golden vector in the reference
 *  and no OpenCV to run.  SmoothTimeSeries and the control flow are the
 * reference's own code. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ---- OpenCV pieces ---- */

/* cv::getGaussianKernel(n, sigma, CV_64F) for sigma > 0 */
static double* cv_gaussian_kernel(int n, double sigma)
{
    double* cd = (double*)malloc(sizeof(double) * (size_t)n);
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX), sum = 0;
    int i;
    for (i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2X * x * x);
        cd[i] = t;
        sum += cd[i];
    }
    sum = 1. / sum;
    for (i = 0; i < n; i++) cd[i] *= sum;
    return cd;
}

/* cv::sepFilter2D(src rows x cols CV_64F, kernelX of kn taps, kernelY = {ky}, anchor centre,
 * delta 0, BORDER_REPLICATE): row pass into a padded line, then the one-tap column pass. */
static void cv_sep_filter_rows(const double* src, int rows, int cols, const double* kx, int kn, double ky, double* dst)
{
    int r, i, k, pad = kn / 2;
    double* line = (double*)malloc(sizeof(double) * (size_t)(cols + kn));
    for (r = 0; r < rows; r++) {
        const double* s = src + (size_t)r * cols;
        for (i = 0; i < cols + kn - 1; i++) {
            int j = i - pad;
            line[i] = s[j < 0 ? 0 : j >= cols ? cols - 1 : j];
        }
        for (i = 0; i < cols; i++) {
            const double* S = line + i;
            double s0 = kx[0] * S[0];
            for (k = 1; k < kn; k++) s0 += kx[k] * S[k];
            dst[(size_t)r * cols + i] = ky * s0 + 0.0;      /* SymmColumnFilter, ksize 1: f*S[0] + delta */
        }
    }
    free(line);
}

/* cv::reduce(src rows x cols, dim = 1, CV_REDUCE_AVG) */
static void cv_reduce_avg_cols(const double* src, int rows, int cols, double* dst)
{
    int y, i;
    for (y = 0; y < rows; y++) {
        const double* s = src + (size_t)y * cols;
        double a0, a1;
        if (cols == 1) { dst[y] = s[0] * (1. / cols) + 0.0; continue; }
        a0 = s[0]; a1 = s[1];
        for (i = 2; i <= cols - 4; i += 4) {
            a0 = a0 + s[i];
            a1 = a1 + s[i + 1];
            a0 = a0 + s[i + 2];
            a1 = a1 + s[i + 3];
        }
        for (; i < cols; i++) a0 = a0 + s[i];
        a0 = a0 + a1;
        dst[y] = a0 * (1. / cols) + 0.0;                    /* convertTo(dst, type, 1./cols) */
    }
}

/* cv::mulTransposed(src rows x cols, dst, aTa = false, delta = rows x 1, scale) = MulTransposedL */
static void cv_mul_transposed_l(const double* src, int rows, int cols, const double* delta, double scale, double* dst)
{
    int i, j, k;
    double* row_buf = (double*)malloc(sizeof(double) * (size_t)cols);
    for (i = 0; i < rows; i++) {
        const double* tsrc1 = src + (size_t)i * cols;
        for (k = 0; k < cols; k++) row_buf[k] = tsrc1[k] - delta[i];
        for (j = i; j < rows; j++) {
            const double* tsrc2 = src + (size_t)j * cols;
            double d = delta[j], s = 0;
            for (k = 0; k <= cols - 4; k += 4)
                s += row_buf[k] * (tsrc2[k] - d) + row_buf[k + 1] * (tsrc2[k + 1] - d) +
                     row_buf[k + 2] * (tsrc2[k + 2] - d) + row_buf[k + 3] * (tsrc2[k + 3] - d);
            for (; k < cols; k++) s += row_buf[k] * (tsrc2[k] - d);
            dst[i * rows + j] = s * scale;
        }
    }
    for (i = 0; i < rows; i++) for (j = 0; j < i; j++) dst[i * rows + j] = dst[j * rows + i];
    free(row_buf);
}

/* cv::eigen on a symmetric n x n matrix: JacobiImpl_<double> (lapack.cpp) */
static void cv_jacobi(double* A, int n, double* W, double* V)
{
    const double eps = DBL_EPSILON;
    int i, j, k, m, iters, maxIters = n * n * 30;
    int* indR = (int*)malloc(sizeof(int) * 2 * (size_t)n);
    int* indC = indR + n;
    double mv;
    for (i = 0; i < n; i++) { for (j = 0; j < n; j++) V[i * n + j] = 0; V[i * n + i] = 1; }
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(A[n * k + m]), i = k + 2; i < n; i++) {
                double val = fabs(A[n * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) {
                double val = fabs(A[n * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    if (n > 1) for (iters = 0; iters < maxIters; iters++) {
        int l;
        double p, y, t, s, c, a0, b0;
        for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
            double val = fabs(A[n * i + indR[i]]);
            if (mv < val) mv = val, k = i;
        }
        l = indR[k];
        for (i = 1; i < n; i++) {
            double val = fabs(A[n * indC[i] + i]);
            if (mv < val) mv = val, k = indC[i], l = i;
        }
        p = A[n * k + l];
        if (fabs(p) <= eps) break;
        y = (W[l] - W[k]) * 0.5;
        t = fabs(y) + hypot(p, y);
        s = hypot(p, t);
        c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] -= t;
        W[l] += t;
#define ROTATE(v0, v1) (a0 = (v0), b0 = (v1), (v0) = a0 * c - b0 * s, (v1) = a0 * s + b0 * c)
        for (i = 0; i < k; i++) ROTATE(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; i++) ROTATE(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; i++) ROTATE(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; i++) ROTATE(V[n * k + i], V[n * l + i]);
#undef ROTATE
        for (j = 0; j < 2; j++) {
            int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx + 1, mv = fabs(A[n * idx + m]), i = idx + 2; i < n; i++) {
                    double val = fabs(A[n * idx + i]);
                    if (mv < val) mv = val, m = i;
                }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) {
                    double val = fabs(A[n * i + idx]);
                    if (mv < val) mv = val, m = i;
                }
                indC[idx] = m;
            }
        }
    }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) {
            double tmp = W[m]; W[m] = W[k]; W[k] = tmp;
            for (i = 0; i < n; i++) { tmp = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = tmp; }
        }
    }
    free(indR);
}

/* cv::gemm(A m x n, B n x p) -> D m x p, alpha 1 (GEMMSingleMul: s0 = 0; s0 += a*b for k in order; s0 *= alpha) */
static void cv_gemm(const double* A, int m, int n, const double* B, int p, double* D)
{
    int i, j, k;
    for (i = 0; i < m; i++) for (j = 0; j < p; j++) {
        double s0 = 0;
        for (k = 0; k < n; k++) s0 += A[i * n + k] * B[k * p + j];
        D[i * p + j] = s0 * 1.0;
    }
}

/* ---- the reference's functions ---- */

/* smoothing.cc:11-47.  q = [n][4] (w, x, y, z), in place.  Returns 0, or -1 for the CHECK_GT(sigma, 0). */
int porc_smooth_heading_directions(double* q, int n, int sigma)
{
    double *kx, *ky, *raw, *smooth;
    int i, c;
    if (sigma <= 0) return -1;
    if (n == 0) return 0;
    kx = cv_gaussian_kernel(sigma * 4 + 1, sigma);
    ky = cv_gaussian_kernel(1, 1);
    raw = (double*)malloc(sizeof(double) * 4 * (size_t)n);
    smooth = (double*)malloc(sizeof(double) * 4 * (size_t)n);
    for (i = 0; i < n; i++) for (c = 0; c < 4; c++) raw[(size_t)c * n + i] = q[4 * (size_t)i + c];   /* raw_rotations(4, N) */
    cv_sep_filter_rows(raw, 4, n, kx, sigma * 4 + 1, ky[0], smooth);
    for (i = 0; i < n; i++) {
        double result = 0, element_norm;
        for (c = 0; c < 4; c++) { double v = smooth[(size_t)c * n + i]; result += v * v; }          /* norm(col, NORM_L2) */
        element_norm = sqrt(result);
        for (c = 0; c < 4; c++) q[4 * (size_t)i + c] = smooth[(size_t)c * n + i] / element_norm;
    }
    free(kx); free(ky); free(raw); free(smooth);
    return 0;
}

static double normal_cdf(double x, double mean, double sigma)       /* smoothing.cc:50-54 */
{
    const double sqrt_2 = sqrt(2.0);
    return 0.5 * (1.0 + erf((x - mean) / (sqrt_2 * sigma)));
}

/* smoothing.cc:57-97 */
int porc_smooth_time_series(const double* values, const double* times, int n, const double* targets, int m,
                            double sigma, double* result)
{
    size_t left_idx = 0, right_idx = 0, integral_idx, N = (size_t)n;
    int target_idx;
    if (!(sigma > 0)) return -1;
    for (target_idx = 0; target_idx < m; ++target_idx) {
        const double target_time = targets[target_idx];
        double prev_point_gaussian_cdf = 0;
        result[target_idx] = 0;
        while (left_idx + 1 < N && (target_time - times[left_idx + 1]) > 3 * sigma) ++left_idx;
        while (right_idx + 1 < N && (times[right_idx] - target_time) < 3 * sigma) ++right_idx;
        for (integral_idx = left_idx; integral_idx < right_idx; ++integral_idx) {
            const double next_timestamp_midpoint = (times[integral_idx] + times[integral_idx + 1]) / 2.0;
            const double next_gaussian_cdf = normal_cdf(next_timestamp_midpoint, target_time, sigma);
            result[target_idx] += values[integral_idx] * (next_gaussian_cdf - prev_point_gaussian_cdf);
            prev_point_gaussian_cdf = next_gaussian_cdf;
        }
        result[target_idx] += values[right_idx] * (1.0 - prev_point_gaussian_cdf);
    }
    return 0;
}

/* track_image_sequence.cc:16-29: cv::PCA(trajectory_matrix(3, N), noArray(), CV_PCA_DATA_AS_COL), N >= 3 */
int porc_trajectory_pca(const double* translations, int n, double* eigenvectors, double* eigenvalues, double* mean)
{
    double* M;
    double covar[9];
    int i, r;
    if (n < 3) return -1;
    M = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    for (i = 0; i < n; i++) for (r = 0; r < 3; r++) M[(size_t)r * n + i] = translations[3 * (size_t)i + r];
    cv_reduce_avg_cols(M, 3, n, mean);
    cv_mul_transposed_l(M, 3, n, mean, 1. / n, covar);
    cv_jacobi(covar, 3, eigenvalues, eigenvectors);
    free(M);
    return 0;
}

/* horizontal_flatten.cc:7-30 */
void porc_project_directions(const double* q, int n, const double* plane, double* dirs)
{
    int i;
    for (i = 0; i < n; i++) {
        const double w = q[4 * (size_t)i], x = q[4 * (size_t)i + 1], y = q[4 * (size_t)i + 2], z = q[4 * (size_t)i + 3];
        const double vx = 0, vy = 0, vz = 1;
        /* Eigen: Vector3 uv = vec().cross(v); uv += uv; return v + w() * uv + vec().cross(uv); */
        double ux = y * vz - z * vy, uy = z * vx - x * vz, uz = x * vy - y * vx;
        double cx, cy, cz, d[3];
        ux += ux; uy += uy; uz += uz;
        cx = y * uz - z * uy; cy = z * ux - x * uz; cz = x * uy - y * ux;
        d[0] = (vx + w * ux) + cx; d[1] = (vy + w * uy) + cy; d[2] = (vz + w * uz) + cz;
        cv_gemm(plane, 2, 3, d, 1, dirs + 2 * (size_t)i);
    }
}

/* horizontal_flatten.cc:32-43 */
void porc_project_translations(double* translations, int n, const double* plane)
{
    int i;
    for (i = 0; i < n; i++) {
        double p[2], o[3];
        cv_gemm(plane, 2, 3, translations + 3 * (size_t)i, 1, p);     /* plane * t : 2x1 */
        cv_gemm(p, 1, 2, plane, 3, o);                               /* (.)^T * plane : 1x3 */
        memcpy(translations + 3 * (size_t)i, o, sizeof(o));
    }
}

/* horizontal_flatten.cc:45-63 */
void porc_turn_angles(const double* dirs, int n, double* turn_angles)
{
    int point_idx, k;
    for (point_idx = 0; point_idx < n; point_idx++) turn_angles[point_idx] = 0;
    for (point_idx = 1; point_idx < n; ++point_idx) {
        const double prev[3] = {dirs[2 * (size_t)(point_idx - 1)], dirs[2 * (size_t)(point_idx - 1) + 1], 0};
        const double curr[3] = {dirs[2 * (size_t)point_idx], dirs[2 * (size_t)point_idx + 1], 0};
        double dot = 0, n2p = 0, n2c = 0, rotation_cos, cross_z;
        for (k = 0; k < 3; k++) dot += prev[k] * curr[k];
        for (k = 0; k < 3; k++) n2p += prev[k] * prev[k];
        for (k = 0; k < 3; k++) n2c += curr[k] * curr[k];
        rotation_cos = dot / sqrt(n2p) / sqrt(n2c);
        cross_z = prev[0] * curr[1] - prev[1] * curr[0];
        turn_angles[point_idx] = acos(rotation_cos) * (cross_z > 0 ? 1.0 : -1.0);
    }
}

/* ---- src/calibration/rotation.cc ---- */

/* cv::PCA(data m x 3, noArray(), CV_PCA_DATA_AS_ROW): reduce(dim 0, AVG), mulTransposed(aTa, delta = mean row), eigen */
static void cv_pca_rows3(const double* data, int m, double* eigenvalues, double* eigenvectors)
{
    double buf[3], mean[3], covar[9];
    double* col_buf = (double*)malloc(sizeof(double) * (size_t)m);
    int i, j, k;
    for (i = 0; i < 3; i++) buf[i] = data[i];                      /* reduceR_: first row, then row after row */
    for (k = 1; k < m; k++) for (i = 0; i < 3; i++) buf[i] = buf[i] + data[3 * (size_t)k + i];
    for (i = 0; i < 3; i++) mean[i] = buf[i] * (1. / m) + 0.0;
    for (i = 0; i < 3; i++) {                                      /* MulTransposedR, width 3: only the scalar tail */
        for (k = 0; k < m; k++) col_buf[k] = data[3 * (size_t)k + i] - mean[i];
        for (j = i; j < 3; j++) {
            double s0 = 0;
            for (k = 0; k < m; k++) s0 += col_buf[k] * (data[3 * (size_t)k + j] - mean[j]);
            covar[3 * i + j] = s0 * (1. / m);
        }
    }
    for (i = 0; i < 3; i++) for (j = 0; j < i; j++) covar[3 * i + j] = covar[3 * j + i];
    cv_jacobi(covar, 3, eigenvalues, eigenvectors);
    free(col_buf);
}

/* rotation.cc:16-57.  Returns 0, -1 for the CHECK_GT on the interval, -2 for fewer than 3 integrated rotations. */
int porc_principal_rotation_axes(const double* rot, const long long* time_usec, int n, long long integration_interval_usec, double* eigenvectors)
{
    double* interval_rotations;
    double q[4] = {1, 0, 0, 0}, eigenvalues[3];                    /* w x y z */
    long long current_interval_usec = 0;
    int rotation_idx, m = 0;
    if (integration_interval_usec <= 0) return -1;
    interval_rotations = (double*)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    for (rotation_idx = 1; rotation_idx < n; ++rotation_idx) {
        const double* r = rot + 3 * (size_t)rotation_idx;
        const long long rotation_duration_usec = time_usec[rotation_idx] - time_usec[rotation_idx - 1];
        const double duration_sec = (double)rotation_duration_usec * 1e-6;
        /* RotationMotionToQuaternion, geometry.cc:6-22 */
        const double rate = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        const double half_theta = rate * duration_sec * 0.5;
        const double sn = sin(half_theta) / (rate + 1e-30);
        const double b[4] = {cos(half_theta), r[0] * sn, r[1] * sn, r[2] * sn};
        double c[4];
        current_interval_usec += rotation_duration_usec;
        c[0] = q[0] * b[0] - q[1] * b[1] - q[2] * b[2] - q[3] * b[3];      /* Eigen quaternion product, generic form */
        c[1] = q[0] * b[1] + q[1] * b[0] + q[2] * b[3] - q[3] * b[2];
        c[2] = q[0] * b[2] + q[2] * b[0] + q[3] * b[1] - q[1] * b[3];
        c[3] = q[0] * b[3] + q[3] * b[0] + q[1] * b[2] - q[2] * b[1];
        memcpy(q, c, sizeof(q));
        if (current_interval_usec >= integration_interval_usec) {
            interval_rotations[3 * m] = q[1]; interval_rotations[3 * m + 1] = q[2]; interval_rotations[3 * m + 2] = q[3]; m++;
            q[0] = 1; q[1] = q[2] = q[3] = 0;
            current_interval_usec = 0;
        }
    }
    if (m < 3) { free(interval_rotations); return -2; }
    cv_pca_rows3(interval_rotations, m, eigenvalues, eigenvectors);
    free(interval_rotations);
    return 0;
}

/* rotation.cc:111-129 */
int porc_angular_velocities_around_axis(const double* rot, int n, const double* axis, double* result)
{
    double s = 0, axis_norm;
    int i, k;
    for (k = 0; k < 3; k++) s += axis[k] * axis[k];
    axis_norm = sqrt(s);
    if (!(axis_norm > 1.0 - 1e-2) || !(axis_norm < 1.0 + 1e-2)) return -1;
    for (i = 0; i < n; i++) {
        double dot = 0;
        for (k = 0; k < 3; k++) dot += rot[3 * (size_t)i + k] * axis[k];
        result[i] = dot / axis_norm;
    }
    return 0;
}

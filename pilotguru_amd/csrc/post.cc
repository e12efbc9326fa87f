// post.cc -- what optical_trajectories does to a finished trajectory before it writes the JSON
// (SURVEY.md §8 row f4).  Host code, double precision, once per segment: it is not a GPU path
// in the reference and it is not one here.  Linked into libpgorb.so so that the CLI and the
// Python tests reach it through the same C ABI as the kernels.
//
// Restates, operation for operation (the summation orders are part of the contract because the
// results are printed with 15 significant digits):
//   SmoothHeadingDirections            src/slam/smoothing.cc:11-47
//     cv::getGaussianKernel(4s+1, s)   OpenCV 2.4 smooth.cpp  (exp, running sum, scale by 1/sum)
//     cv::sepFilter2D(64F, REPLICATE)  OpenCV 2.4 filter.cpp  RowFilter<double,double>: k = 0..n-1 in order;
//                                      the 1-tap column pass is v*1.0 + 0.0
//     cv::norm(col, NORM_L2)           sqrt(((w^2 + x^2) + y^2) + z^2)
//   SmoothTimeSeries                   src/slam/smoothing.cc:49-97
//   TrajectoryToPCA                    src/slam/track_image_sequence.cc:16-29
//     cv::PCA(DATA_AS_COL)             OpenCV 2.4 matmul.cpp: reduce(AVG) with two alternating partial sums,
//                                      MulTransposedL (groups of four products), scale 1/n, cv::eigen = JacobiImpl_
//   GetPrincipalRotationAxes, GetAngularVelocitiesAroundAxisDirect   src/calibration/rotation.cc:16-57,111-129
//     cv::PCA(DATA_AS_ROW)             reduce over rows = one running sum per column, MulTransposedR = one running
//                                      sum per matrix element (3 columns: the 4-wide unrolled path is not taken)
//   ProjectDirections / ProjectTranslations / Projected2DDirectionsToTurnAngles
//                                      src/slam/horizontal_flatten.cc:7-63 (Eigen's _transformVector,
//                                      cv::gemm's k-ordered dot products, cv::Vec dot / cross / norm)
// The OpenCV orders are written from memory of the 2.4 sources (the image has no OpenCV): see
// DESIGN.md "parity unpinned".
#include <cmath>
#include <cstdint>
#include <limits>
#include <utility>
#include <vector>
#include "../../include/pgorb.h"

namespace {

std::vector<double> gaussian_kernel(int n, double sigma)
{
    std::vector<double> k(n);
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        k[i] = std::exp(scale2X * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (double& v : k) v *= sum;
    return k;
}

// one row through RowFilter + the single-tap column filter, BORDER_REPLICATE
void filter_row(const double* src, int n, const std::vector<double>& kx, double ky0, double* dst)
{
    const int ks = (int)kx.size(), anchor = ks / 2;
    auto at = [&](int i) { return src[i < 0 ? 0 : (i >= n ? n - 1 : i)]; };
    for (int i = 0; i < n; i++) {
        double s = kx[0] * at(i - anchor);
        for (int k = 1; k < ks; k++) s += kx[k] * at(i - anchor + k);
        dst[i] = ky0 * s + 0.0;
    }
}

struct Jacobi3 {
    // cv::eigen for a symmetric n x n double matrix (n = 3 here, written for any n)
    static void run(std::vector<double>& A, int n, std::vector<double>& W, std::vector<double>& V)
    {
        const double eps = std::numeric_limits<double>::epsilon();
        W.assign(n, 0); V.assign((size_t)n * n, 0);
        for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1;
        std::vector<int> indR(n), indC(n);
        auto a = [&](int r, int c) -> double& { return A[(size_t)r * n + c]; };
        auto scan_row = [&](int k) {
            int m = k + 1; double mv = std::abs(a(k, m));
            for (int i = k + 2; i < n; i++) { const double v = std::abs(a(k, i)); if (mv < v) mv = v, m = i; }
            indR[k] = m;
        };
        auto scan_col = [&](int k) {
            int m = 0; double mv = std::abs(a(0, k));
            for (int i = 1; i < k; i++) { const double v = std::abs(a(i, k)); if (mv < v) mv = v, m = i; }
            indC[k] = m;
        };
        for (int k = 0; k < n; k++) {
            W[k] = a(k, k);
            if (k < n - 1) scan_row(k);
            if (k > 0) scan_col(k);
        }
        if (n > 1) for (int iters = 0, maxIters = n * n * 30; iters < maxIters; iters++) {
            int k = 0; double mv = std::abs(a(0, indR[0]));
            for (int i = 1; i < n - 1; i++) { const double v = std::abs(a(i, indR[i])); if (mv < v) mv = v, k = i; }
            int l = indR[k];
            for (int i = 1; i < n; i++) { const double v = std::abs(a(indC[i], i)); if (mv < v) mv = v, k = indC[i], l = i; }
            const double p = a(k, l);
            if (std::abs(p) <= eps) break;
            const double y = (W[l] - W[k]) * 0.5;
            double t = std::abs(y) + hypot(p, y);
            double s = hypot(p, t);
            const double c = t / s;
            s = p / s; t = (p / t) * p;
            if (y < 0) s = -s, t = -t;
            a(k, l) = 0;
            W[k] -= t; W[l] += t;
            auto rot = [&](double& v0, double& v1) { const double a0 = v0, b0 = v1; v0 = a0 * c - b0 * s; v1 = a0 * s + b0 * c; };
            for (int i = 0; i < k; i++) rot(a(i, k), a(i, l));
            for (int i = k + 1; i < l; i++) rot(a(k, i), a(i, l));
            for (int i = l + 1; i < n; i++) rot(a(k, i), a(l, i));
            for (int i = 0; i < n; i++) rot(V[(size_t)k * n + i], V[(size_t)l * n + i]);
            for (int j = 0; j < 2; j++) {
                const int idx = j == 0 ? k : l;
                if (idx < n - 1) scan_row(idx);
                if (idx > 0) scan_col(idx);
            }
        }
        for (int k = 0; k < n - 1; k++) {            // eigenvalues descending, rows of V with them
            int m = k;
            for (int i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
            if (k != m) { std::swap(W[m], W[k]); for (int i = 0; i < n; i++) std::swap(V[(size_t)m * n + i], V[(size_t)k * n + i]); }
        }
    }
};

inline double normal_cdf(double x, double mean, double sigma)
{
    static const double sqrt_2 = std::sqrt(2.0);
    return 0.5 * (1.0 + std::erf((x - mean) / (sqrt_2 * sigma)));
}

}  // namespace

extern "C" {

int pgorb_smooth_heading_directions(double* quat_wxyz, int n, int sigma)
{
    if (!quat_wxyz || n < 0) return PGORB_E_ARG;
    if (sigma <= 0) return PGORB_E_ARG;                       // CHECK_GT(sigma, 0), smoothing.cc:14
    if (n == 0) return PGORB_OK;
    const std::vector<double> kx = gaussian_kernel(sigma * 4 + 1, sigma);
    const double ky0 = gaussian_kernel(1, 1)[0];
    std::vector<double> raw((size_t)4 * n), smooth((size_t)4 * n);
    for (int i = 0; i < n; i++) for (int c = 0; c < 4; c++) raw[(size_t)c * n + i] = quat_wxyz[4 * (size_t)i + c];
    for (int c = 0; c < 4; c++) filter_row(&raw[(size_t)c * n], n, kx, ky0, &smooth[(size_t)c * n]);
    for (int i = 0; i < n; i++) {
        double s = 0;
        for (int c = 0; c < 4; c++) { const double v = smooth[(size_t)c * n + i]; s += v * v; }
        const double norm = std::sqrt(s);
        for (int c = 0; c < 4; c++) quat_wxyz[4 * (size_t)i + c] = smooth[(size_t)c * n + i] / norm;
    }
    return PGORB_OK;
}

int pgorb_smooth_time_series(const double* values, const double* times, int n,
                             const double* targets, int m, double sigma, double* out)
{
    if (n < 0 || m < 0 || (n && (!values || !times)) || (m && (!targets || !out))) return PGORB_E_ARG;
    if (!(sigma > 0)) return PGORB_E_ARG;                     // CHECK_GT(sigma, 0), smoothing.cc:63
    if (m && n == 0) return PGORB_E_ARG;                      // data_values.at(right_idx) would throw
    size_t left = 0, right = 0;
    const size_t N = (size_t)n;
    for (int t = 0; t < m; t++) {
        const double target = targets[t];
        while (left + 1 < N && (target - times[left + 1]) > 3 * sigma) ++left;
        while (right + 1 < N && (times[right] - target) < 3 * sigma) ++right;
        double prev_cdf = 0, acc = 0;
        for (size_t i = left; i < right; ++i) {
            const double mid = (times[i] + times[i + 1]) / 2.0;
            const double cdf = normal_cdf(mid, target, sigma);
            acc += values[i] * (cdf - prev_cdf);
            prev_cdf = cdf;
        }
        acc += values[right] * (1.0 - prev_cdf);
        out[t] = acc;
    }
    return PGORB_OK;
}

int pgorb_trajectory_pca(const double* translations, int n, double* eigenvectors, double* eigenvalues, double* mean)
{
    if (!translations || !eigenvectors || !eigenvalues) return PGORB_E_ARG;
    if (n < 3) return PGORB_E_LIMIT;          // cv::PCA switches to the "scrambled" covariance below 3 samples
    double mu[3];
    for (int r = 0; r < 3; r++) {                             // reduce(..., CV_REDUCE_AVG): two alternating partial sums
        auto src = [&](int i) { return translations[3 * (size_t)i + r]; };
        double a0 = src(0), a1 = src(1);
        int i = 2;
        for (; i <= n - 4; i += 4) { a0 = a0 + src(i); a1 = a1 + src(i + 1); a0 = a0 + src(i + 2); a1 = a1 + src(i + 3); }
        for (; i < n; i++) a0 = a0 + src(i);
        a0 = a0 + a1;
        mu[r] = a0 * (1. / n) + 0.0;
    }
    std::vector<double> cov(9), rowbuf(n);
    const double scale = 1. / n;
    for (int i = 0; i < 3; i++) {                             // MulTransposedL with a one-column delta
        for (int k = 0; k < n; k++) rowbuf[k] = translations[3 * (size_t)k + i] - mu[i];
        for (int j = i; j < 3; j++) {
            auto d = [&](int k) { return translations[3 * (size_t)k + j] - mu[j]; };
            double s = 0;
            int k = 0;
            for (; k <= n - 4; k += 4)
                s += rowbuf[k] * d(k) + rowbuf[k + 1] * d(k + 1) + rowbuf[k + 2] * d(k + 2) + rowbuf[k + 3] * d(k + 3);
            for (; k < n; k++) s += rowbuf[k] * d(k);
            cov[3 * i + j] = s * scale;
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < i; j++) cov[3 * i + j] = cov[3 * j + i];    // completeSymm
    std::vector<double> W, V;
    Jacobi3::run(cov, 3, W, V);
    for (int i = 0; i < 3; i++) eigenvalues[i] = W[i];
    for (int i = 0; i < 9; i++) eigenvectors[i] = V[i];
    if (mean) for (int i = 0; i < 3; i++) mean[i] = mu[i];
    return PGORB_OK;
}

int pgorb_project_directions(const double* quat_wxyz, int n, const double* plane, double* dirs)
{
    if (n < 0 || (n && (!quat_wxyz || !dirs)) || !plane) return PGORB_E_ARG;
    const double v[3] = {0, 0, 1};
    for (int i = 0; i < n; i++) {
        const double w = quat_wxyz[4 * (size_t)i], q[3] = {quat_wxyz[4 * (size_t)i + 1], quat_wxyz[4 * (size_t)i + 2], quat_wxyz[4 * (size_t)i + 3]};
        // Eigen QuaternionBase::_transformVector: uv = vec x v; uv += uv; v + w*uv + vec x uv
        double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
        for (double& u : uv) u += u;
        const double cr[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
        double d[3];
        for (int c = 0; c < 3; c++) d[c] = (v[c] + w * uv[c]) + cr[c];
        for (int r = 0; r < 2; r++) {                         // cv::gemm: k-ordered dot product, times alpha = 1
            double s = 0;
            for (int k = 0; k < 3; k++) s += plane[3 * r + k] * d[k];
            dirs[2 * (size_t)i + r] = s * 1.0;
        }
    }
    return PGORB_OK;
}

int pgorb_project_translations(double* translations, int n, const double* plane)
{
    if (n < 0 || (n && !translations) || !plane) return PGORB_E_ARG;
    for (int i = 0; i < n; i++) {
        double* t = translations + 3 * (size_t)i;
        double p[2];
        for (int r = 0; r < 2; r++) { double s = 0; for (int k = 0; k < 3; k++) s += plane[3 * r + k] * t[k]; p[r] = s * 1.0; }
        double o[3];
        for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 2; k++) s += p[k] * plane[3 * k + c]; o[c] = s * 1.0; }
        for (int c = 0; c < 3; c++) t[c] = o[c];
    }
    return PGORB_OK;
}

int pgorb_turn_angles(const double* dirs, int n, double* turn)
{
    if (n < 0 || (n && (!dirs || !turn))) return PGORB_E_ARG;
    if (n) turn[0] = 0;
    for (int i = 1; i < n; i++) {
        const double prev[3] = {dirs[2 * (size_t)i - 2], dirs[2 * (size_t)i - 1], 0}, curr[3] = {dirs[2 * (size_t)i], dirs[2 * (size_t)i + 1], 0};
        double dot = 0, np = 0, nc = 0;
        for (int k = 0; k < 3; k++) { dot += prev[k] * curr[k]; np += prev[k] * prev[k]; nc += curr[k] * curr[k]; }
        const double rotation_cos = dot / std::sqrt(np) / std::sqrt(nc);
        const double cross_z = prev[0] * curr[1] - prev[1] * curr[0];
        turn[i] = std::acos(rotation_cos) * (cross_z > 0 ? 1.0 : -1.0);
    }
    return PGORB_OK;
}

int pgorb_principal_rotation_axes(const double* rot, const int64_t* time_usec, int n, int64_t integration_interval_usec,
                                  double* eigenvectors)
{
    if (!rot || !time_usec || !eigenvectors || n < 0) return PGORB_E_ARG;
    if (integration_interval_usec <= 0) return PGORB_E_ARG;                 // CHECK_GT, rotation.cc:19
    std::vector<double> rows;                                               // interval_rotations, x y z per row
    double cw = 1, cx = 0, cy = 0, cz = 0;
    int64_t cur_usec = 0;
    for (int i = 1; i < n; i++) {
        const int64_t dur = time_usec[i] - time_usec[i - 1];
        cur_usec += dur;
        const double rx = rot[3 * (size_t)i], ry = rot[3 * (size_t)i + 1], rz = rot[3 * (size_t)i + 2];
        const double duration_sec = static_cast<double>(dur) * 1e-6;
        const double rate = std::sqrt(rx * rx + ry * ry + rz * rz);            // RotationMotionToQuaternion, geometry.cc:6-22
        const double half_theta = rate * duration_sec * 0.5;
        const double sn = std::sin(half_theta) / (rate + 1e-30);
        const double bw = std::cos(half_theta), bx = rx * sn, by = ry * sn, bz = rz * sn;
        const double nw = cw * bw - cx * bx - cy * by - cz * bz, nx = cw * bx + cx * bw + cy * bz - cz * by;
        const double ny = cw * by + cy * bw + cz * bx - cx * bz, nz = cw * bz + cz * bw + cx * by - cy * bx;
        cw = nw; cx = nx; cy = ny; cz = nz;
        if (cur_usec >= integration_interval_usec) {
            rows.push_back(cx); rows.push_back(cy); rows.push_back(cz);
            cw = 1; cx = cy = cz = 0;
            cur_usec = 0;
        }
    }
    const int m = (int)(rows.size() / 3);
    if (m < 3) return PGORB_E_LIMIT;                                        // CHECK_GE(interval_rotations.size(), 3), :47
    double mu[3];
    for (int j = 0; j < 3; j++) {                                           // reduce(dim 0, AVG)
        double b = rows[j];
        for (int k = 1; k < m; k++) b = b + rows[3 * (size_t)k + j];
        mu[j] = b * (1. / m) + 0.0;
    }
    std::vector<double> cov(9), col(m);
    for (int i = 0; i < 3; i++) {                                           // MulTransposedR, delta = the mean row
        for (int k = 0; k < m; k++) col[k] = rows[3 * (size_t)k + i] - mu[i];
        for (int j = i; j < 3; j++) {
            double s0 = 0;
            for (int k = 0; k < m; k++) s0 += col[k] * (rows[3 * (size_t)k + j] - mu[j]);
            cov[3 * i + j] = s0 * (1. / m);
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < i; j++) cov[3 * i + j] = cov[3 * j + i];
    std::vector<double> W, V;
    Jacobi3::run(cov, 3, W, V);
    for (int i = 0; i < 9; i++) eigenvectors[i] = V[i];
    return PGORB_OK;
}

int pgorb_angular_velocities_around_axis(const double* rot, int n, const double* axis, double* out)
{
    if (n < 0 || (n && (!rot || !out)) || !axis) return PGORB_E_ARG;
    double s = 0;
    for (int k = 0; k < 3; k++) s += axis[k] * axis[k];
    const double axis_norm = std::sqrt(s);
    if (!(axis_norm > 1.0 - 1e-2) || !(axis_norm < 1.0 + 1e-2)) return PGORB_E_ARG;    // the two CHECKs, rotation.cc:114-116
    for (int i = 0; i < n; i++) {
        double d = 0;
        for (int k = 0; k < 3; k++) d += rot[3 * (size_t)i + k] * axis[k];
        out[i] = d / axis_norm;
    }
    return PGORB_OK;
}

int pgorb_kahan_sum(const double* values, int n, int dim, double* sum)
{
    if (n < 0 || dim <= 0 || !sum || (n && !values)) return PGORB_E_ARG;
    for (int k = 0; k < dim; k++) {
        double s = 0.0, rem = 0.0;
        for (int i = 0; i < n; i++) {                          // KahanSum::add, include/math/math.hpp:13-19
            const double proposed = values[(size_t)i * dim + k] + rem;
            const double updated = s + proposed;
            const double actual = updated - s;
            rem = proposed - actual;
            s = updated;
        }
        sum[k] = s;
    }
    return PGORB_OK;
}

}  // extern "C"

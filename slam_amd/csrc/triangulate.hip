// Linear (DLT) triangulation, one lane per correspondence, everything in registers.
//
// Replaces linear_least_square + triangulate (utils/utills.py:220-255): A = [x_l P2 - P0; y_l P2 - P1;
// x_r Q2 - Q0; y_r Q2 - Q1] (4x4, float64), right singular vector of the least singular value,
// X = v[:3] / v[3] (no cheirality or range rejection, as in the reference).  The reference calls LAPACK gesdd per
// point.  Only ONE singular vector is wanted, the one of the smallest singular value, so no full SVD is computed:
//   1. A = Q R by three Householder reflections (Q is never formed: A and R have the same right singular vectors);
//   2. inverse iteration on R^T R with the two triangular solves done on R itself, x <- R^-1 R^-T x, normalised -- the
//      condition number is not squared the way forming A^T A would square it; started from R^-1 e4.
// The iteration converges linearly with ratio (s4 / s3)^2.  For a DLT matrix s3 is set by the rig (>= 263 for the KITTI
// calibration whatever the point) and s4 by the detection noise (s4 / s3: median 8e-4, at most 0.06 on the bench data),
// so 3-5 iterations settle the vector to rounding; they stop when an iteration moves no component by more than 1e-12
// (the error is then below ratio x 1e-12), capped at 24 for matrices without such a gap, whose vector is not defined
// to that accuracy by any method.  Against a 40-digit SVD the result is as close as LAPACK's own or closer (median
// relative deviation of X 2e-15 against 1e-14; tools/triangulation_accuracy.py).
// Round 1-2 ran a one-sided Jacobi SVD of the whole 4x4 here: ~2600 FP64 instructions per point (5-6 sweeps of 480);
// this takes ~450 (profiles/r3_*: 0.50 -> see DESIGN section 4).
#include "common.h"
#include "jacobi.h"   // rsqrt_nr, rcp_nr
#include "kernels.h"

namespace {

constexpr int TRI_MAX_ITER = 24;

__device__ __forceinline__ void dlt_point(const Mat34 &P, const Mat34 &Q, double xl, double yl, double xr, double yr,
                                          double *out) {
    double A[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[0][j] = xl * P.m[8 + j] - P.m[j];
        A[1][j] = yl * P.m[8 + j] - P.m[4 + j];
        A[2][j] = xr * Q.m[8 + j] - Q.m[j];
        A[3][j] = yr * Q.m[8 + j] - Q.m[4 + j];
    }
    // ---- Householder QR, R left in the upper triangle of A ----
    double amax = 0;                         // scale of the matrix, for the pivot guard below
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double n2 = 0;
#pragma unroll
        for (int i = k; i < 4; ++i) n2 = fma(A[i][k], A[i][k], n2);
        const double alpha = n2 > 0.0 ? n2 * rsqrt_nr(n2) : 0.0;        // |column k below the diagonal, inclusive|
        const double x0 = A[k][k];
        const double v0 = x0 + copysign(alpha, x0);                     // no cancellation
        // H = I - beta v v^T with v = (v0, A[k+1..][k]); v^T v = 2 alpha (alpha + |x0|)
        const double den = alpha * (alpha + fabs(x0));
        const double beta = den > 0.0 ? rcp_nr(den) : 0.0;
#pragma unroll
        for (int j = k + 1; j < 4; ++j) {
            double w = v0 * A[k][j];
#pragma unroll
            for (int i = k + 1; i < 4; ++i) w = fma(A[i][k], A[i][j], w);
            w *= beta;
            A[k][j] = fma(-w, v0, A[k][j]);
#pragma unroll
            for (int i = k + 1; i < 4; ++i) A[i][j] = fma(-w, A[i][k], A[i][j]);
        }
        A[k][k] = -copysign(alpha, x0);
        amax = fmax(amax, alpha);
    }
    amax = fmax(amax, fabs(A[3][3]));
    // reciprocal pivots; a pivot that vanished (exact data: s4 = 0) is replaced by a rounding-level one, which sends the
    // iteration straight to the null vector
    const double tiny = amax * 2.220446049250313e-16;
    double d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double r = A[i][i];
        d[i] = rcp_nr(fabs(r) > tiny ? r : (tiny > 0.0 ? tiny : 1.0));
    }
    // ---- inverse iteration on R^T R ----
    double x[4] = {0.0, 0.0, 0.0, 1.0};
    for (int it = 0; it < TRI_MAX_ITER; ++it) {
        // R^T w = x (forward), R z = w (backward)
        const double w0 = x[0] * d[0];
        const double w1 = fma(-A[0][1], w0, x[1]) * d[1];
        const double w2 = fma(-A[1][2], w1, fma(-A[0][2], w0, x[2])) * d[2];
        const double w3 = fma(-A[2][3], w2, fma(-A[1][3], w1, fma(-A[0][3], w0, x[3]))) * d[3];
        double z[4];
        z[3] = w3 * d[3];
        z[2] = fma(-A[2][3], z[3], w2) * d[2];
        z[1] = fma(-A[1][3], z[3], fma(-A[1][2], z[2], w1)) * d[1];
        z[0] = fma(-A[0][3], z[3], fma(-A[0][2], z[2], fma(-A[0][1], z[1], w0))) * d[0];
        const double n2 = fma(z[0], z[0], fma(z[1], z[1], fma(z[2], z[2], z[3] * z[3])));
        // (R^T R)^-1 is positive definite: z . x > 0, so successive iterates keep their sign
        const double inv = rsqrt_nr(n2);
        double moved = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double zi = z[i] * inv;
            moved = fmax(moved, fabs(zi - x[i]));
            x[i] = zi;
        }
        if (moved <= 1e-12) break;           // NaN / inf inputs never pass and run to the cap
    }
    out[0] = x[0] / x[3];
    out[1] = x[1] / x[3];
    out[2] = x[2] / x[3];
}

__global__ __launch_bounds__(256) void triangulate_kernel(Mat34 P, Mat34 Q, const float4 *xy, const int *count,
                                                          int seg_stride, double *X) {
    const int s = blockIdx.y;
    const int n = count ? count[s] : seg_stride;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t g = (size_t)s * seg_stride + i;
    const float4 c = xy[g];
    double out[3];
    dlt_point(P, Q, (double)c.x, (double)c.y, (double)c.z, (double)c.w, out);
    X[g * 3 + 0] = out[0];
    X[g * 3 + 1] = out[1];
    X[g * 3 + 2] = out[2];
}

__global__ __launch_bounds__(256) void triangulate_f64_kernel(Mat34 P, Mat34 Q, const double2 *l, const double2 *r,
                                                              int n, double *X) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 a = l[i], b = r[i];
    double out[3];
    dlt_point(P, Q, a.x, a.y, b.x, b.y, out);
    X[(size_t)i * 3 + 0] = out[0];
    X[(size_t)i * 3 + 1] = out[1];
    X[(size_t)i * 3 + 2] = out[2];
}

Mat34 load34(const double *h) {
    Mat34 m;
    for (int i = 0; i < 12; ++i) m.m[i] = h[i];
    return m;
}

}  // namespace

int slam_launch_triangulate(const Mat34 &P, const Mat34 &Q, const float *d_xy, const int32_t *d_count, int n_seg,
                            int seg_stride, double *d_X, hipStream_t st) {
    if (n_seg <= 0 || seg_stride <= 0) return SLAM_OK;
    SLAM_REQUIRE(d_xy && d_X, "slam_triangulate: null pointer");
    SLAM_REQUIRE(n_seg <= 65535, "slam_triangulate: more than 65535 segments per call");
    hipLaunchKernelGGL(triangulate_kernel, dim3(slam_div_up(seg_stride, 256), n_seg), dim3(256), 0, st, P, Q,
                       (const float4 *)d_xy, d_count, seg_stride, d_X);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_triangulate(const double *h_P, const double *h_Q, const float *d_xy, const int32_t *d_count,
                                int n_seg, int seg_stride, double *d_X, void *stream) {
    SLAM_REQUIRE(h_P && h_Q, "slam_triangulate: null camera matrix");
    return slam_launch_triangulate(load34(h_P), load34(h_Q), d_xy, d_count, n_seg, seg_stride, d_X,
                                   (hipStream_t)stream);
}

extern "C" int slam_triangulate_f64(const double *h_P, const double *h_Q, const double *d_xy_l, const double *d_xy_r,
                                    int n, double *d_X, void *stream) {
    if (n <= 0) return SLAM_OK;
    SLAM_REQUIRE(h_P && h_Q && d_xy_l && d_xy_r && d_X, "slam_triangulate_f64: null pointer");
    hipLaunchKernelGGL(triangulate_f64_kernel, dim3(slam_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       load34(h_P), load34(h_Q), (const double2 *)d_xy_l, (const double2 *)d_xy_r, n, d_X);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

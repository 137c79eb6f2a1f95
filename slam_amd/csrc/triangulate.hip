// Linear (DLT) triangulation, one lane per correspondence, everything in registers.
//
// Replaces linear_least_square + triangulate (utils/utills.py:220-255): A = [x_l P2 - P0; y_l P2 - P1;
// x_r Q2 - Q0; y_r Q2 - Q1] (4x4, float64), right singular vector of the least singular value,
// X = v[:3] / v[3].  The reference calls LAPACK gesdd per point; here a one-sided Jacobi SVD runs
// on the 4x4 in registers (high relative accuracy, no cheirality or range rejection, as in the reference).
#include "common.h"
#include "jacobi.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ void dlt_point(const Mat34 &P, const Mat34 &Q, double xl, double yl, double xr, double yr,
                                          double *out) {
    double A[4][4], V[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[0][j] = xl * P.m[8 + j] - P.m[j];
        A[1][j] = yl * P.m[8 + j] - P.m[4 + j];
        A[2][j] = xr * Q.m[8 + j] - Q.m[j];
        A[3][j] = yr * Q.m[8 + j] - Q.m[4 + j];
    }
    jacobi4(A, V);
    double best = 0;
    int k = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double n2 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) n2 = fma(A[i][j], A[i][j], n2);
        if (j == 0 || n2 < best) {
            best = n2;
            k = j;
        }
    }
    double v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = k == 0 ? V[i][0] : k == 1 ? V[i][1] : k == 2 ? V[i][2] : V[i][3];
    out[0] = v[0] / v[3];
    out[1] = v[1] / v[3];
    out[2] = v[2] / v[3];
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

// One-sided (Hestenes) Jacobi SVD of a small square matrix held in registers.
// Columns of A are rotated until mutually orthogonal: A_out = A_in * V, singular values = column norms.
#pragma once
#include <hip/hip_runtime.h>

template <int N>
__device__ __forceinline__ void jacobi_onesided(double (&A)[N][N], double (&V)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < N - 1; ++p) {
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    alpha = fma(A[i][p], A[i][p], alpha);
                    beta = fma(A[i][q], A[i][q], beta);
                    gamma = fma(A[i][p], A[i][q], gamma);
                }
                if (gamma != 0.0 && fabs(gamma) > 1e-16 * sqrt(alpha * beta)) {
                    rotated = true;
                    const double zeta = (beta - alpha) / (2.0 * gamma);
                    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t);
                    const double s = c * t;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const double ap = A[i][p], aq = A[i][q];
                        A[i][p] = c * ap - s * aq;
                        A[i][q] = s * ap + c * aq;
                        const double vp = V[i][p], vq = V[i][q];
                        V[i][p] = c * vp - s * vq;
                        V[i][q] = s * vp + c * vq;
                    }
                }
            }
        }
        if (!rotated) break;
    }
}

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
                if (gamma != 0.0 && fabs(gamma) > 1e-15 * sqrt(alpha * beta)) {
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

// ---- 4x4 specialisation for the DLT triangulation -------------------------------------------------------------
// Same algorithm (Hestenes one-sided Jacobi with V accumulation), arranged for throughput: each sweep is three
// steps of two DISJOINT column pairs, (0,1)(2,3), (0,2)(1,3), (0,3)(1,2), so two independent scalar chains
// (2 square roots, 2 divisions each) are in flight per step, and a converged pair is a rotation by zero rather
// than a branch.
__device__ __forceinline__ void jacobi_rot_params(double alpha, double beta, double gamma, double &c, double &s,
                                                  bool &rotated) {
    // converged when |gamma| <= 1e-15 sqrt(alpha beta), tested without the square root.  (At 1e-16 about 1.5 % of
    // DLT matrices keep rotating at rounding level until the sweep cap, and one such lane stalls its whole wave.)
    const bool go = gamma * gamma > 1e-30 * (alpha * beta);
    const double d = beta - alpha;
    const double den = fabs(d) + sqrt(fma(d, d, 4.0 * gamma * gamma));
    double t = (2.0 * gamma) / den;
    t = d < 0.0 ? -t : t;
    t = go ? t : 0.0;                      // also absorbs 0/0 for a zero column pair
    c = 1.0 / sqrt(fma(t, t, 1.0));      // not rsqrt(): the device rsqrt is not accurate to double precision
    s = c * t;
    rotated |= go;
}

template <int P, int Q>
__device__ __forceinline__ void jacobi_apply(double (&A)[4][4], double (&V)[4][4], double c, double s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double ap = A[i][P], aq = A[i][Q];
        A[i][P] = c * ap - s * aq;
        A[i][Q] = s * ap + c * aq;
        const double vp = V[i][P], vq = V[i][Q];
        V[i][P] = c * vp - s * vq;
        V[i][Q] = s * vp + c * vq;
    }
}

template <int P, int Q, int R, int S>
__device__ __forceinline__ void jacobi_step(double (&A)[4][4], double (&V)[4][4], bool &rotated) {
    double a1 = 0, b1 = 0, g1 = 0, a2 = 0, b2 = 0, g2 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a1 = fma(A[i][P], A[i][P], a1);
        b1 = fma(A[i][Q], A[i][Q], b1);
        g1 = fma(A[i][P], A[i][Q], g1);
        a2 = fma(A[i][R], A[i][R], a2);
        b2 = fma(A[i][S], A[i][S], b2);
        g2 = fma(A[i][R], A[i][S], g2);
    }
    double c1, s1, c2, s2;
    jacobi_rot_params(a1, b1, g1, c1, s1, rotated);
    jacobi_rot_params(a2, b2, g2, c2, s2, rotated);
    jacobi_apply<P, Q>(A, V, c1, s1);
    jacobi_apply<R, S>(A, V, c2, s2);
}

__device__ __forceinline__ void jacobi4(double (&A)[4][4], double (&V)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        jacobi_step<0, 1, 2, 3>(A, V, rotated);
        jacobi_step<0, 2, 1, 3>(A, V, rotated);
        jacobi_step<0, 3, 1, 2>(A, V, rotated);
        if (!rotated) break;
    }
}

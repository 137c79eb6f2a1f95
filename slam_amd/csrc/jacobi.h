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
// 1/sqrt(x) and 1/x to double accuracy from the hardware estimates (v_rsq_f64 / v_rcp_f64, ~2^-26) and two Newton
// steps each: ~10 instructions where the library sqrt() and division cost ~25 and ~15 with their range fix-ups.  The
// arguments here are sums of squares of pixel-scale numbers and 1 + t^2, far from the ranges those fix-ups are for.
// A rotation angle that is off by an ulp only changes how fast the columns become orthogonal, not where they end up.
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    return y;
}

__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

__device__ __forceinline__ void jacobi_rot_params(double alpha, double beta, double gamma, double &c, double &s,
                                                  bool &rotated) {
    // converged when |gamma| <= 1e-15 sqrt(alpha beta), tested without the square root.  (At 1e-16 about 1.5 % of
    // DLT matrices keep rotating at rounding level until the sweep cap, and one such lane stalls its whole wave.)
    const double g2 = gamma * gamma, ab = alpha * beta;
    const bool go = g2 > 1e-30 * ab;
    const double d = beta - alpha;
    const double q = fma(d, d, 4.0 * gamma * gamma);
    const double den = fabs(d) + q * rsqrt_nr(q);          // q = 0 gives NaN here; then go is false and t is discarded
    double t = (2.0 * gamma) * rcp_nr(den);
    t = d < 0.0 ? -t : t;
    t = go ? t : 0.0;                      // also absorbs 0/0 for a zero column pair
    c = rsqrt_nr(fma(t, t, 1.0));
    s = c * t;
    // Convergence is quadratic: a pair whose cosine is below 1e-9 before this rotation is orthogonal to rounding after
    // it, so only larger ones ask for another sweep.  (Asking for one after every rotation costs a last sweep that
    // rotates nothing: 0.66 -> 0.59 ms per shard, points bit-identical on the bench data.)
    rotated |= g2 > 1e-18 * ab;
}

// The rotation with one product fused (2 operations per entry instead of 3; the library is compiled with
// -ffp-contract=off, so the fusion is spelled out).  Nothing downstream depends on the last bit of a rotated column.
template <int P, int Q>
__device__ __forceinline__ void jacobi_apply(double (&A)[4][4], double (&V)[4][4], double c, double s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double ap = A[i][P], aq = A[i][Q];
        A[i][P] = fma(c, ap, -(s * aq));
        A[i][Q] = fma(s, ap, c * aq);
        const double vp = V[i][P], vq = V[i][Q];
        V[i][P] = fma(c, vp, -(s * vq));
        V[i][Q] = fma(s, vp, c * vq);
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

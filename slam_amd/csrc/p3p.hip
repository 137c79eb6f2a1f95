// Minimal-sample mapping and AP3P hypothesis generation, one lane per hypothesis.
//
// Replaces pnp(world[idx], left1[idx], K, cv2.SOLVEPNP_AP3P) + rodriguez_to_mat
// (utils/utills.py:574-596, called at :503) for every RANSAC draw at once.
//
// The arithmetic of cv2.solvePnP(AP3P) is OpenCV's ap3p.cpp (Ke & Roumeliotis, CVPR 2017): bearing vectors of
// the first three points, a quartic in cos(theta1'), Ferrari solve + two Newton polish sweeps, up to four poses,
// ordered by the fourth point's reprojection error (stable, strict '>'), first one returned; then the Python
// side converts R -> rvec -> R through cv2.Rodrigues (range check, SVD orthonormalisation).  OpenCV is not
// vendored by the reference and not installed here: this restates the published algorithm ("parity unpinned"),
// and oracle/slam_oracle.py restates it independently in numpy.
#include "common.h"
#include "jacobi.h"
#include "kernels.h"

namespace {

// ---- the sample stream: np.random.seed(seed); np.random.randint(0, n, 4) x draws ----
// numpy's bounded integers are masked rejection on 32-bit MT19937 outputs (legacy generator).
// One wavefront per row regenerates the pair's MT19937 stream from its seed (init_genrand, as numpy's legacy seeding of
// an integer) and writes draws [begin, begin + n_draw) of the masked-rejection sequence.  Round 0 asks for draws [0, D0),
// the later rounds for ranges that depend on how far each pair's loop has got; regenerating from the seed costs a
// few microseconds per thousand draws and keeps no per-pair generator state between rounds.
// The 624-word state lives in LDS.  The twist is the sequential recurrence evaluated 64 words at a time: word k needs
// the OLD words k and k + 1 and, for k < 227, the old word k + 397, else the NEW word k - 227 -- all available when a
// chunk reads before it writes and chunks run in ascending order.
__global__ __launch_bounds__(64) void draw_samples_kernel(const uint32_t *seed, const int *count, const int *seg_of,
                                                          const int *begin, int n_draw, int *sample, int *produced) {
    __shared__ uint32_t mt[624];
    const int row = blockIdx.x, lane = threadIdx.x;
    const int seg = seg_of ? seg_of[row] : row;
    const int n = count[seg];
    int *out = sample + (size_t)row * n_draw * 4;
    const long want0 = begin ? (long)begin[row] * 4 : 0, want1 = want0 + (long)n_draw * 4;
    if (produced && lane == 0) produced[row] = n > 0 ? n_draw : 0;
    if (n <= 0) return;
    if (n == 1) {   // rng == 0: numpy returns `low` without consuming the stream
        for (int i = lane; i < n_draw * 4; i += 64) out[i] = 0;
        return;
    }
    const uint32_t rng = (uint32_t)n - 1u;
    uint32_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    if (lane == 0) {
        uint32_t v = seed[seg];
        for (int i = 0; i < 624; ++i) {
            mt[i] = v;
            v = 1812433253u * (v ^ (v >> 30)) + (uint32_t)i + 1u;
        }
    }
    __syncthreads();
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
    long got = 0;
    while (got < want1) {
        for (int k0 = 0; k0 < 623; k0 += 64) {
            const int k = k0 + lane;
            uint32_t v = 0;
            if (k < 623) {
                const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO);
                v = mt[k < 227 ? k + 397 : k - 227] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
            }
            __syncthreads();                 // every read of this chunk before any of its writes
            if (k < 623) mt[k] = v;
            __syncthreads();
        }
        if (lane == 0) {
            const uint32_t y = (mt[623] & UP) | (mt[0] & LO);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        __syncthreads();
        for (int k0 = 0; k0 < 624 && got < want1; k0 += 64) {
            const int k = k0 + lane;
            uint32_t v = 0;
            bool acc = false;
            if (k < 624) {
                uint32_t y = mt[k];
                y ^= y >> 11;
                y ^= (y << 7) & 0x9d2c5680u;
                y ^= (y << 15) & 0xefc60000u;
                y ^= y >> 18;
                v = y & mask;
                acc = v <= rng;
            }
            const unsigned long long bal = __ballot(acc);
            const long pos = got + __popcll(bal & ((1ull << lane) - 1ull));
            if (acc && pos >= want0 && pos < want1) out[pos - want0] = (int)v;
            got += __popcll(bal);
        }
        __syncthreads();
    }
}

// ---- complex helpers (principal branches, as <complex>) ----
struct cplx { double re, im; };

__device__ __forceinline__ cplx c_sqrt(cplx z) {
    if (z.re == 0.0 && z.im == 0.0) return {0.0, z.im};
    const double r = hypot(z.re, z.im);
    const double s = sqrt(0.5 * (r + fabs(z.re)));
    if (z.re >= 0.0) return {s, z.im / (2.0 * s)};
    return {fabs(z.im) / (2.0 * s), copysign(s, z.im)};
}

__device__ __forceinline__ cplx c_div_real(double a, cplx b) {   // a / b, Smith's algorithm
    if (fabs(b.re) >= fabs(b.im)) {
        const double r = b.im / b.re, den = b.re + b.im * r;
        return {a / den, (-a * r) / den};
    }
    const double r = b.re / b.im, den = b.re * r + b.im;
    return {(a * r) / den, -a / den};
}

__device__ void solve_quartic(const double *f, double *roots) {
    const double a4 = f[0], a3 = f[1], a2 = f[2], a1 = f[3], a0 = f[4];
    const double a4_2 = a4 * a4, a3_2 = a3 * a3, a4_3 = a4_2 * a4, a2a4 = a2 * a4;
    const double p4 = (8 * a2a4 - 3 * a3_2) / (8 * a4_2);
    const double q4 = (a3_2 * a3 - 4 * a2a4 * a3 + 8 * a1 * a4_2) / (8 * a4_3);
    const double r4 = (256 * a0 * a4_3 - 3 * (a3_2 * a3_2) - 64 * a1 * a3 * a4_2 + 16 * a2a4 * a3_2) / (256 * (a4_3 * a4));
    const double p3 = ((p4 * p4) / 12 + r4) / 3;
    const double q3 = (72 * r4 * p4 - 2 * p4 * p4 * p4 - 27 * q4 * q4) / 432;
    cplx w = c_sqrt({q3 * q3 - p3 * p3 * p3, 0.0});
    if (q3 >= 0) {
        w.re = -w.re - q3;
        w.im = -w.im;
    } else {
        w.re = w.re - q3;
    }
    double t;
    if (w.im == 0.0) {
        const double wr = cbrt(w.re);
        t = 2.0 * (wr + p3 / wr);
    } else {
        const double rho = hypot(w.re, w.im), th = atan2(w.im, w.re);
        t = 4.0 * (cbrt(rho) * cos(th / 3.0));
    }
    const cplx sqrt_2m = c_sqrt({-2 * p4 / 3 + t, 0.0});
    const double B_4A = -a3 / (4 * a4);
    const double complex1 = 4 * p4 / 3 + t;
    const cplx complex2 = c_div_real(2 * q4, sqrt_2m);
    const double sqrt_2m_rh = sqrt_2m.re / 2;
    const double sqrt1 = c_sqrt({-(complex1 + complex2.re), -complex2.im}).re / 2;
    const double sqrt2 = c_sqrt({-(complex1 - complex2.re), complex2.im}).re / 2;
    roots[0] = B_4A + sqrt_2m_rh + sqrt1;
    roots[1] = B_4A + sqrt_2m_rh - sqrt1;
    roots[2] = B_4A - sqrt_2m_rh + sqrt2;
    roots[3] = B_4A - sqrt_2m_rh - sqrt2;
}

__device__ __forceinline__ void cross3(const double *a, const double *b, double *r) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double norm3(const double *a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

__device__ __forceinline__ void matmul3(const double (&a)[3][3], const double (&b)[3][3], double (&r)[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

// cv2.Rodrigues(R) -> rvec -> cv2.Rodrigues(rvec) -> R, the round trip the reference's pnp() performs
__device__ void rodrigues_round_trip(double (&R)[3][3]) {
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) bad |= !(R[i][j] >= -100.0 && R[i][j] < 100.0);   // also catches NaN
    double r[3] = {0, 0, 0};
    if (!bad) {
        // nearest rotation: R = U W Vt -> U Vt, via one-sided Jacobi (R V = U W)
        double A[3][3], V[3][3], U[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) A[i][j] = R[i][j];
        jacobi_onesided<3>(A, V);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double n = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
#pragma unroll
            for (int i = 0; i < 3; ++i) U[i][j] = A[i][j] / n;
        }
        double Q[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Q[i][j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + U[i][2] * V[j][2];
        double rx = Q[2][1] - Q[1][2], ry = Q[0][2] - Q[2][0], rz = Q[1][0] - Q[0][1];
        const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
        double c = (Q[0][0] + Q[1][1] + Q[2][2] - 1) * 0.5;
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        double theta = acos(c);
        if (s < 1e-5) {
            if (c > 0) {
                rx = ry = rz = 0;
            } else {
                double t = (Q[0][0] + 1) * 0.5;
                rx = sqrt(fmax(t, 0.0));
                t = (Q[1][1] + 1) * 0.5;
                ry = sqrt(fmax(t, 0.0)) * (Q[0][1] < 0 ? -1.0 : 1.0);
                t = (Q[2][2] + 1) * 0.5;
                rz = sqrt(fmax(t, 0.0)) * (Q[0][2] < 0 ? -1.0 : 1.0);
                if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && ((Q[1][2] > 0) != (ry * rz > 0))) rz = -rz;
                theta /= sqrt(rx * rx + ry * ry + rz * rz);
                rx *= theta;
                ry *= theta;
                rz *= theta;
            }
        } else {
            double vth = 1.0 / (2.0 * s);
            vth *= theta;
            rx *= vth;
            ry *= vth;
            rz *= vth;
        }
        r[0] = rx;
        r[1] = ry;
        r[2] = rz;
    }
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (!(theta >= 2.220446049250313e-16)) {
        const bool nan = theta != theta;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) R[i][j] = nan ? theta : (i == j ? 1.0 : 0.0);
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1.0 - c;
    const double k[3] = {r[0] / theta, r[1] / theta, r[2] / theta};
    const double kx[3][3] = {{0, -k[2], k[1]}, {k[2], 0, -k[0]}, {-k[1], k[0], 0}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = c * (i == j ? 1.0 : 0.0) + c1 * (k[i] * k[j]) + s * kx[i][j];
}

// returns the number of AP3P solutions; pose = first solution after the stable sort on the 4th point's error
__device__ int ap3p_pose(const Mat33 &K, const double (&X)[4][3], const double (&uv)[4][2], double *pose) {
    const double fx = K.m[0], fy = K.m[4], cx = K.m[2], cy = K.m[5];
    const double inv_fx = 1.0 / fx, inv_fy = 1.0 / fy, cx_fx = cx / fx, cy_fy = cy / fy;
    double b[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double mu = inv_fx * uv[i][0] - cx_fx, mv = inv_fy * uv[i][1] - cy_fy;
        const double mk = 1.0 / sqrt(mu * mu + mv * mv + 1);
        b[i][0] = mu * mk;
        b[i][1] = mv * mk;
        b[i][2] = mk;
    }
    const double mu3 = inv_fx * uv[3][0] - cx_fx, mv3 = inv_fy * uv[3][1] - cy_fy;
    const double *w1 = X[0], *w2 = X[1], *w3 = X[2], *w4 = X[3];
    double u0[3] = {w1[0] - w2[0], w1[1] - w2[1], w1[2] - w2[2]};
    const double nu0 = norm3(u0);
    double k1[3] = {u0[0] / nu0, u0[1] / nu0, u0[2] / nu0};
    double k3[3];
    cross3(b[0], b[1], k3);
    const double nk3 = norm3(k3);
    k3[0] /= nk3;
    k3[1] /= nk3;
    k3[2] /= nk3;
    double tz[3], v1[3], v2[3];
    cross3(b[0], k3, tz);
    cross3(b[0], b[2], v1);
    cross3(b[1], b[2], v2);
    double u1[3] = {w1[0] - w3[0], w1[1] - w3[1], w1[2] - w3[2]};
    const double u1k1 = dot3(u1, k1), k3b3 = dot3(k3, b[2]);
    double f11 = k3b3, f13 = dot3(k3, v1);
    const double f15 = -u1k1 * f11;
    double nl[3];
    cross3(u1, k1, nl);
    const double delta = norm3(nl);
    nl[0] /= delta;
    nl[1] /= delta;
    nl[2] /= delta;
    f11 *= delta;
    f13 *= delta;
    const double u2k1 = u1k1 - nu0;
    double f21 = dot3(tz, v2), f22 = nk3 * k3b3, f23 = dot3(k3, v2);
    const double f24 = u2k1 * f22, f25 = -u2k1 * f21;
    f21 *= delta;
    f22 *= delta;
    f23 *= delta;
    const double g1 = f13 * f22, g2 = f13 * f25 - f15 * f23, g3 = f11 * f23 - f13 * f21, g4 = -f13 * f24;
    const double g5 = f11 * f22, g6 = f11 * f25 - f15 * f21, g7 = -f15 * f24;
    const double coeffs[5] = {g5 * g5 + g1 * g1 + g3 * g3, 2 * (g5 * g6 + g1 * g2 + g3 * g4),
                              g6 * g6 + 2 * g5 * g7 + g2 * g2 + g4 * g4 - g1 * g1 - g3 * g3,
                              2 * (g6 * g7 - g1 * g2 - g3 * g4), g7 * g7 - g2 * g2 - g4 * g4};
    double s[4];
    solve_quartic(coeffs, s);
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double r = s[j];
            const double err = (((coeffs[0] * r + coeffs[1]) * r + coeffs[2]) * r + coeffs[3]) * r + coeffs[4];
            const double der = ((4 * coeffs[0] * r + 3 * coeffs[1]) * r + 2 * coeffs[2]) * r + coeffs[3];
            s[j] = r - err / der;
        }
    double temp[3];
    cross3(k1, nl, temp);
    const double Ck1nl[3][3] = {{k1[0], nl[0], temp[0]}, {k1[1], nl[1], temp[1]}, {k1[2], nl[2], temp[2]}};
    const double Cb[3][3] = {{b[0][0], b[0][1], b[0][2]}, {k3[0], k3[1], k3[2]}, {tz[0], tz[1], tz[2]}};
    const double sc = delta / k3b3;
    const double b3p[3] = {sc * b[2][0], sc * b[2][1], sc * b[2][2]};

    int n_sol = 0;
    int order[4] = {0, 1, 2, 3};
    double errs[4];
    double sol[4][12];
    for (int i = 0; i < 4; ++i) {
        const double ct1 = s[i];
        if (fabs(ct1) > 1) continue;            // NaN falls through, as in the C++
        double st1 = sqrt(1 - ct1 * ct1);
        st1 = (k3b3 > 0) ? st1 : -st1;
        double ct3 = g1 * ct1 + g2, st3 = g3 * ct1 + g4;
        const double nt3 = st1 / ((g5 * ct1 + g6) * ct1 + g7);
        ct3 *= nt3;
        st3 *= nt3;
        const double C13[3][3] = {{ct3, 0, -st3}, {st1 * st3, ct1, st1 * ct3}, {ct1 * st3, -st1, ct1 * ct3}};
        double tmp[3][3], R[3][3];
        matmul3(Ck1nl, C13, tmp);
        matmul3(tmp, Cb, R);
        double *o = sol[n_sol];
        // solution rotation = R^T, translation = st1 * b3p - R^T w3
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double rp3 = w3[0] * R[0][a] + w3[1] * R[1][a] + w3[2] * R[2][a];
            o[a * 4 + 0] = R[0][a];
            o[a * 4 + 1] = R[1][a];
            o[a * 4 + 2] = R[2][a];
            o[a * 4 + 3] = st1 * b3p[a] - rp3;
        }
        const double Xp = o[0] * w4[0] + o[1] * w4[1] + o[2] * w4[2] + o[3];
        const double Yp = o[4] * w4[0] + o[5] * w4[1] + o[6] * w4[2] + o[7];
        const double Zp = o[8] * w4[0] + o[9] * w4[1] + o[10] * w4[2] + o[11];
        const double mu3p = Xp / Zp, mv3p = Yp / Zp;
        errs[n_sol] = (mu3p - mu3) * (mu3p - mu3) + (mv3p - mv3) * (mv3p - mv3);
        ++n_sol;
    }
    if (n_sol == 0) return 0;
    // stable insertion sort on strict '>' (NaN never swaps), tracking only the permutation
    for (int i = 1; i < n_sol; ++i)
        for (int j = i; j > 0 && errs[order[j - 1]] > errs[order[j]]; --j) {
            const int t = order[j];
            order[j] = order[j - 1];
            order[j - 1] = t;
        }
    const int w = order[0];
    double R[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[a][c] = w == 0 ? sol[0][a * 4 + c] : w == 1 ? sol[1][a * 4 + c] : w == 2 ? sol[2][a * 4 + c] : sol[3][a * 4 + c];
    rodrigues_round_trip(R);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        pose[a * 4 + 0] = R[a][0];
        pose[a * 4 + 1] = R[a][1];
        pose[a * 4 + 2] = R[a][2];
        pose[a * 4 + 3] = w == 0 ? sol[0][a * 4 + 3] : w == 1 ? sol[1][a * 4 + 3] : w == 2 ? sol[2][a * 4 + 3] : sol[3][a * 4 + 3];
    }
    return n_sol;
}

template <typename XY4>
__global__ __launch_bounds__(64) void p3p_kernel(Mat33 K, const double *X, const XY4 *xy1, const int *count,
                                                 const int *seg_of, int seg_stride, const int *sample,
                                                 int sample_stride, int draw0, int n_hyp, double *pose,
                                                 uint8_t *status) {
    const int row = blockIdx.y;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_hyp) return;
    const int seg = seg_of ? seg_of[row] : row;
    const int n = count[seg];
    const size_t o = (size_t)row * n_hyp + h;
    double *out = pose + o * 12;
    if (n <= 0) {
        status[o] = 0;
        for (int i = 0; i < 12; ++i) out[i] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    const int *idx = sample + ((size_t)row * sample_stride + (size_t)(draw0 + h)) * 4;
    double Xs[4][3], uv[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int i = idx[k];
        i = i < 0 ? 0 : (i >= n ? n - 1 : i);       // indices come from draw_samples and are in range; clamp defensively
        const size_t g = (size_t)seg * seg_stride + i;
        Xs[k][0] = X[g * 3 + 0];
        Xs[k][1] = X[g * 3 + 1];
        Xs[k][2] = X[g * 3 + 2];
        const XY4 c = xy1[g];
        uv[k][0] = (double)c.x;
        uv[k][1] = (double)c.y;
    }
    double T[12];
    const int ns = ap3p_pose(K, Xs, uv, T);
    status[o] = ns > 0 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) out[i] = ns > 0 ? T[i] : __longlong_as_double(0x7ff8000000000000ll);
}

}  // namespace

int slam_launch_draw_samples(const uint32_t *d_seed, const int32_t *d_count, const int32_t *d_seg_of,
                             const int32_t *d_begin, int n_rows, int n_draw, int32_t *d_sample, int32_t *d_produced,
                             hipStream_t st) {
    if (n_rows <= 0 || n_draw <= 0) return SLAM_OK;
    hipLaunchKernelGGL(draw_samples_kernel, dim3(n_rows), dim3(64), 0, st, d_seed, d_count, d_seg_of, d_begin, n_draw,
                       d_sample, d_produced);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

int slam_launch_p3p(const Mat33 &K, const double *d_X, const void *d_xy1, int xy_f64, const int32_t *d_count,
                    const int32_t *d_seg_of, int n_rows, int seg_stride, const int32_t *d_sample, int sample_stride,
                    int draw0, int n_hyp, double *d_pose, uint8_t *d_status, hipStream_t st) {
    if (n_rows <= 0 || n_hyp <= 0) return SLAM_OK;
    SLAM_REQUIRE(n_rows <= 65535, "slam_p3p_hypotheses: more than 65535 segments per call");
    if (xy_f64)
        hipLaunchKernelGGL(p3p_kernel<double4>, dim3(slam_div_up(n_hyp, 64), n_rows), dim3(64), 0, st, K, d_X,
                           (const double4 *)d_xy1, d_count, d_seg_of, seg_stride, d_sample, sample_stride, draw0,
                           n_hyp, d_pose, d_status);
    else
        hipLaunchKernelGGL(p3p_kernel<float4>, dim3(slam_div_up(n_hyp, 64), n_rows), dim3(64), 0, st, K, d_X,
                           (const float4 *)d_xy1, d_count, d_seg_of, seg_stride, d_sample, sample_stride, draw0,
                           n_hyp, d_pose, d_status);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_randint_draws(const uint32_t *d_seed, const int32_t *d_count, const int32_t *d_begin, int n_rows,
                                  int n_draw, int32_t *d_sample, void *stream) {
    SLAM_REQUIRE(n_rows >= 0 && n_draw >= 0, "slam_randint_draws: negative size");
    if (n_rows == 0 || n_draw == 0) return SLAM_OK;
    SLAM_REQUIRE(d_seed && d_count && d_begin && d_sample, "slam_randint_draws: null pointer");
    return slam_launch_draw_samples(d_seed, d_count, nullptr, d_begin, n_rows, n_draw, d_sample, nullptr, (hipStream_t)stream);
}

extern "C" int slam_p3p_hypotheses(const double *h_K, const double *d_X, const void *d_xy1, int xy_f64,
                                   const int32_t *d_count, int n_seg, int seg_stride, const int32_t *d_sample,
                                   int n_hyp, double *d_pose, uint8_t *d_status, void *stream) {
    SLAM_REQUIRE(h_K && d_X && d_xy1 && d_count && d_sample && d_pose && d_status, "slam_p3p_hypotheses: null pointer");
    Mat33 K;
    for (int i = 0; i < 9; ++i) K.m[i] = h_K[i];
    return slam_launch_p3p(K, d_X, d_xy1, xy_f64, d_count, nullptr, n_seg, seg_stride, d_sample, n_hyp, 0, n_hyp, d_pose,
                           d_status, (hipStream_t)stream);
}

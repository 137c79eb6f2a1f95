// Consensus scoring: one wavefront per pose hypothesis (two when only counting), 64 points per step, ballot + popcount.
//
// Replaces consensus_match (utils/utills.py:533-545) -> compose_transformations (:659-669),
// find_supporters_4_frame (:672-695), project (:698-705), check_projection (:708-716) and
// compute_square_dist (:719-736).
//
// The arithmetic follows numpy's operation order so that the supporter masks are the reference's, bit for bit,
// for the same pose matrix:
//   - `calib @ T`, `M2 @ [T; 0 0 0 1]` and `X @ P[:, :3].T` go through BLAS dgemm, whose kernels accumulate
//     k = 0, 1, 2(, 3) with fused multiply-adds from a zero accumulator (checked against numpy/OpenBLAS in the
//     build container): c = fma(a2, b2, fma(a1, b1, a0 * b0)).
//   - `+ P[:, 3]`, the two divisions by the third homogeneous coordinate, `proj - obs`, the products and the sum
//     inside np.linalg.norm, the square root and the comparison are separate correctly rounded operations.
//   - the test is sqrt(dx*dx + dy*dy) <= acc**2 (the reference compares the UN-squared distance with acc squared);
//     no positive-depth test; NaN compares false.
// Compiled with -ffp-contract=off, so nothing below is fused except the explicit fma() calls.
#include <cmath>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

struct Proj { double p[12]; };

// dst(3x4) = A(3x3) * B(3x4), BLAS accumulation order
__device__ __forceinline__ void gemm_33_34(const double *A, const double *B, double *dst) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dst[i * 4 + j] = fma(A[i * 3 + 2], B[8 + j], fma(A[i * 3 + 1], B[4 + j], A[i * 3 + 0] * B[j]));
}

// dst(3x4) = M(3x4) * [T; 0 0 0 1](4x4)
__device__ __forceinline__ void compose_34(const double *M, const double *T, double *dst) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double last = j == 3 ? 1.0 : 0.0;
            dst[i * 4 + j] = fma(M[i * 4 + 3], last, fma(M[i * 4 + 2], T[8 + j], fma(M[i * 4 + 1], T[4 + j], M[i * 4 + 0] * T[j])));
        }
}

__device__ __forceinline__ bool within(const double *P, double x, double y, double z, double u, double v, double thr) {
    const double hx = fma(z, P[2], fma(y, P[1], x * P[0])) + P[3];
    const double hy = fma(z, P[6], fma(y, P[5], x * P[4])) + P[7];
    const double hz = fma(z, P[10], fma(y, P[9], x * P[8])) + P[11];
    const double dx = hx / hz - u, dy = hy / hz - v;
    return sqrt(dx * dx + dy * dy) <= thr;
}

// Fast classification of the same test: 1 = inside, 0 = outside, -1 = too close to the threshold to call.
// The homogeneous coordinates are the exact ones above; the two divisions and the square root are avoided by testing
//     (hx - u hz)^2 + (hy - v hz)^2   against   thr^2 hz^2 (1 -+ 1e-9)
// (the reference's inequality multiplied by hz^2 >= 0; it has no positive-depth test, so the sign of hz does not
// matter).  hx - u hz is ONE fused multiply-add, hence correctly rounded however much cancels, and both sides carry a
// relative error of a few 1e-16: five orders below the band, so every decisive answer equals the exact test's.  The
// undecided point takes the exact path, and so does everything irregular: a NaN or infinity anywhere makes both
// comparisons false; hz = 0 gives thresholds -+TINY, which no square reaches from the right side; TINY also keeps
// a tiny hz (hz^2 underflowing) from deciding anything, since products lose at most ~1e-322 to underflow.
// (r1 used a Newton-refined hardware reciprocal here: 5 more instructions, one of them quarter rate.)
constexpr double TINY = 1e-250;

__device__ __forceinline__ int within_fast(const double *P, double x, double y, double z, double u, double v, double lo,
                                           double hi) {
    const double hx = fma(z, P[2], fma(y, P[1], x * P[0])) + P[3];
    const double hy = fma(z, P[6], fma(y, P[5], x * P[4])) + P[7];
    const double hz = fma(z, P[10], fma(y, P[9], x * P[8])) + P[11];
    const double a = fma(-u, hz, hx), b = fma(-v, hz, hy);       // fused: this path only classifies against the band
    const double q = fma(a, a, b * b), z2 = hz * hz;
    return q < fma(lo, z2, -TINY) ? 1 : (q > fma(hi, z2, TINY) ? 0 : -1);
}

// Both cameras at once for a rectified pair (RECT, decided on the host from K and M2, see rectified_rig): with
// K = [fx 0 cx; 0 fy cy; 0 0 1] and M2 = [I | (tx, 0, 0)] numpy's `M2 @ [T; 0 0 0 1]` returns the rows of T unchanged
// except element (0, 3), and `K @ .` then gives a right projection matrix that differs from the left one in element
// (0, 3) only -- bit for bit, because the extra terms are exact zeros and ones.  So hy, hz and the dot product inside hx
// are the left camera's; only the addend of hx and the observation differ.  Returns the AND of the two classifications:
// 1 inside both, 0 outside one, -1 undecided (then the caller runs the exact test on the full matrices, which also
// covers non-finite poses: a NaN anywhere in T makes a left coordinate NaN, hence q NaN, hence -1).
__device__ __forceinline__ int within_fast_rect(const double *Pl, double pr3, double x, double y, double z, double ul,
                                                double vl, double ur, double vr, double lo, double hi) {
    const double sx = fma(z, Pl[2], fma(y, Pl[1], x * Pl[0]));
    const double hxl = sx + Pl[3], hxr = sx + pr3;
    const double hy = fma(z, Pl[6], fma(y, Pl[5], x * Pl[4])) + Pl[7];
    const double hz = fma(z, Pl[10], fma(y, Pl[9], x * Pl[8])) + Pl[11];
    const double al = fma(-ul, hz, hxl), bl = fma(-vl, hz, hy), ar = fma(-ur, hz, hxr), br = fma(-vr, hz, hy);
    const double ql = fma(al, al, bl * bl), qr = fma(ar, ar, br * br), z2 = hz * hz;
    const double lo2 = fma(lo, z2, -TINY), hi2 = fma(hi, z2, TINY);
    if (ql > hi2 || qr > hi2) return 0;
    return (ql < lo2 && qr < lo2) ? 1 : -1;
}

constexpr int SB = 256;   // 4 waves per workgroup, all on the same segment (shared L1 footprint)

// One wavefront scores HW hypotheses of one segment: every point (24 B + one observation) is loaded once and tested
// against the HW pose pairs held in registers, so the L1/L2 traffic per hypothesis drops by HW.  HW = 2 for counting
// (48 doubles of projection matrices per hypothesis), 1 when a mask row is written.
template <typename XY4, int HW, bool RECT>
__global__ __launch_bounds__(SB) void score_kernel(Mat33 K, Mat34 M2, const double *X, const XY4 *xy1,
                                                   const int *count, const int *seg_of, int n_rows, int groups,
                                                   int seg_stride, const double *pose, const uint8_t *status, int n_hyp,
                                                   double thr, int *inlier_count, unsigned long long *mask,
                                                   const int *gate) {
    if (gate && *gate != 0) return;          // queued speculatively behind a device-side decision that went the other way
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  All `groups` workgroups
    // of a segment (they read the same points) therefore go to one XCD: XCD c takes the segments c, c + 8, ...
    // (With the hypothesis group as blockIdx.x every segment's points were fetched by all eight L2s: 1.35 GB of
    // fabric reads per launch for 0.19 GB of points and poses.)
    const int xcd = blockIdx.x % 8, k = blockIdx.x / 8;
    const int row = (k / groups) * 8 + xcd;
    if (row >= n_rows) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h0 = ((k % groups) * (SB / 64) + wave) * HW;
    if (h0 >= n_hyp) return;
    const int seg = seg_of ? seg_of[row] : row;
    const int n = count[seg];
    const int words = seg_stride / 64;

    double Pl[HW][12], Pr[HW][12];
    bool live[HW];
    bool any_live = false;
#pragma unroll
    for (int k = 0; k < HW; ++k) {
        const int h = h0 + k < n_hyp ? h0 + k : h0;          // a wave past the end re-reads its first hypothesis, unused
        const size_t o = (size_t)row * n_hyp + h;
        live[k] = h0 + k < n_hyp && status[o] != 0;
        any_live |= live[k];
        double T[12], Tr[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = pose[o * 12 + i];
        compose_34(M2.m, T, Tr);
        gemm_33_34(K.m, T, Pl[k]);
        gemm_33_34(K.m, Tr, Pr[k]);
        if (RECT) {
            // the right matrix is the left one except element (0, 3) (see within_fast_rect): keep 13 scalars per
            // hypothesis and rebuild the right matrix from them where the exact test needs it
#pragma unroll
            for (int i = 0; i < 12; ++i) Pl[k][i] = to_scalar(Pl[k][i]);
            const double pr3 = to_scalar(Pr[k][3]);
#pragma unroll
            for (int i = 0; i < 12; ++i) Pr[k][i] = Pl[k][i];
            Pr[k][3] = pr3;
        }
    }

    const double *Xs = X + (size_t)seg * seg_stride * 3;
    const XY4 *obs = xy1 + (size_t)seg * seg_stride;
    // thr < 0 never passes the exact test: an empty band below zero classifies every finite distance as outside
    const double lo = thr < 0.0 ? -1.0 : thr * thr * (1.0 - 1e-9), hi = thr < 0.0 ? -1.0 : thr * thr * (1.0 + 1e-9);
    int total[HW];
#pragma unroll
    for (int k = 0; k < HW; ++k) total[k] = 0;
    unsigned long long *mrow = mask ? mask + ((size_t)row * n_hyp + h0) * words : nullptr;   // HW == 1 when mask != null
    if (any_live || mrow) {
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            bool ok[HW];
#pragma unroll
            for (int k = 0; k < HW; ++k) ok[k] = false;
            if (any_live && i < n) {
                const double x = Xs[i * 3 + 0], y = Xs[i * 3 + 1], z = Xs[i * 3 + 2];
                const XY4 c = obs[i];
                const double ul = (double)c.x, vl = (double)c.y, ur = (double)c.z, vr = (double)c.w;
#pragma unroll
                for (int k = 0; k < HW; ++k) {
                    int f;
                    if (RECT) {
                        f = within_fast_rect(Pl[k], Pr[k][3], x, y, z, ul, vl, ur, vr, lo, hi);
                    } else {
                        const int fl = within_fast(Pl[k], x, y, z, ul, vl, lo, hi);
                        const int fr = within_fast(Pr[k], x, y, z, ur, vr, lo, hi);
                        // decisively outside in one camera: the AND is false either way
                        f = (fl == 0 || fr == 0) ? 0 : ((fl == 1 && fr == 1) ? 1 : -1);
                    }
                    bool in = f == 1;
                    if (f < 0)                  // within 1e-9 of the radius (or NaN): the reference's arithmetic, literally
                        in = within(Pl[k], x, y, z, ul, vl, thr) && within(Pr[k], x, y, z, ur, vr, thr);
                    ok[k] = in && live[k];
                }
            }
#pragma unroll
            for (int k = 0; k < HW; ++k) {
                const unsigned long long bal = __ballot(ok[k]);
                total[k] += __popcll(bal);
                if (HW == 1 && mrow && lane == 0) mrow[base >> 6] = bal;
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < HW; ++k)
            if (h0 + k < n_hyp) inlier_count[(size_t)row * n_hyp + h0 + k] = total[k];
    }
    if (HW == 1 && mrow) {
        for (int w = (n + 63) / 64 + lane; w < words; w += 64) mrow[w] = 0ull;
    }
}

// Counting only, many hypotheses per segment (the later RANSAC rounds: up to 2^20 draws over the pairs whose loop is
// still running): ONE LANE PER HYPOTHESIS.  The wavefront-per-hypothesis kernel above builds every hypothesis's
// projection matrices in all 64 lanes and spends a whole 64-point step on a segment that holds 20 points, which is
// what the pairs that run to the iteration cap look like (unrelated frames in loop-closure validation): ~130 FP64
// wave-instructions per hypothesis whatever the segment size.  Here a lane builds the matrices of its own hypothesis
// once, the segment's points go through LDS in tiles of LT points that every lane reads as broadcasts, and the cost is
// (130 + 30 T) / 64 wave-instructions per hypothesis.  The per-point arithmetic is the same functions as above, so
// the counts are the same bit for bit.
// A matrix with a NaN entry makes the homogeneous coordinate it feeds NaN for every point (fma and add propagate NaN
// whatever the other operand), hence the distance NaN, hence the comparison false: such a hypothesis (a degenerate
// sample, 6/T of the draws) has no supporters and is not evaluated; otherwise its lane would drag the wavefront
// through the exact divide + sqrt path on every point.  Infinite entries take the ordinary path.
constexpr int LT = 256;   // points per LDS tile (7 doubles each: 14 KB)
#ifndef SLAM_LANES_MIN_HYP
#define SLAM_LANES_MIN_HYP 256
#endif
constexpr int LANES_MIN_HYP = SLAM_LANES_MIN_HYP;   // hypotheses per segment from which the lane-per-hypothesis kernel is used

template <typename XY4, bool RECT>
__global__ __launch_bounds__(SB) void score_lanes_kernel(Mat33 K, Mat34 M2, const double *X, const XY4 *xy1,
                                                         const int *count, const int *seg_of, int n_rows, int groups,
                                                         int seg_stride, const double *pose, const uint8_t *status,
                                                         int n_hyp, double thr, int *inlier_count) {
    __shared__ double pts[LT * 7];
    // all workgroups of a segment on one XCD, as in score_kernel
    const int xcd = blockIdx.x % 8, k = blockIdx.x / 8;
    const int row = (k / groups) * 8 + xcd;
    if (row >= n_rows) return;
    const int h = (k % groups) * SB + threadIdx.x;
    const bool have = h < n_hyp;
    const size_t o = (size_t)row * n_hyp + (have ? h : 0);
    const int seg = seg_of ? seg_of[row] : row;
    const int n = count[seg];

    double Pl[12], Pr[12];
    {
        double T[12], Tr[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = pose[o * 12 + i];
        compose_34(M2.m, T, Tr);
        gemm_33_34(K.m, T, Pl);
        gemm_33_34(K.m, Tr, Pr);
    }
    bool live = have && status[o] != 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) live = live && !(Pl[i] != Pl[i]) && !(Pr[i] != Pr[i]);

    const double *Xs = X + (size_t)seg * seg_stride * 3;
    const XY4 *obs = xy1 + (size_t)seg * seg_stride;
    const double lo = thr < 0.0 ? -1.0 : thr * thr * (1.0 - 1e-9), hi = thr < 0.0 ? -1.0 : thr * thr * (1.0 + 1e-9);
    // fast classification of point i of the tile: 1 inside, 0 outside, -1 undecided
    auto classify = [&](int i) -> int {
        const double *q = pts + i * 7;
        if (RECT) return within_fast_rect(Pl, Pr[3], q[0], q[1], q[2], q[3], q[4], q[5], q[6], lo, hi);
        const int fl = within_fast(Pl, q[0], q[1], q[2], q[3], q[4], lo, hi);
        const int fr = within_fast(Pr, q[0], q[1], q[2], q[5], q[6], lo, hi);
        return (fl == 0 || fr == 0) ? 0 : ((fl == 1 && fr == 1) ? 1 : -1);
    };
    // the reference's arithmetic, literally (points within 1e-9 of the radius, non-finite values)
    auto exact = [&](int i) -> bool {
        const double *q = pts + i * 7;
        if (RECT) {
            // the right matrix is the left one except element (0, 3) (see within_fast_rect): 13 live doubles, not 24
            double R[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) R[e] = Pl[e];
            R[3] = Pr[3];
            return within(Pl, q[0], q[1], q[2], q[3], q[4], thr) && within(R, q[0], q[1], q[2], q[5], q[6], thr);
        }
        return within(Pl, q[0], q[1], q[2], q[3], q[4], thr) && within(Pr, q[0], q[1], q[2], q[5], q[6], thr);
    };
    int total = 0;
    for (int base = 0; base < n; base += LT) {
        if (base) __syncthreads();                     // the previous tile has been consumed
        for (int i = threadIdx.x; i < LT && base + i < n; i += SB) {
            const int g = base + i;
            const XY4 c = obs[g];
            double *q = pts + i * 7;
            q[0] = Xs[g * 3 + 0];
            q[1] = Xs[g * 3 + 1];
            q[2] = Xs[g * 3 + 2];
            q[3] = (double)c.x;
            q[4] = (double)c.y;
            q[5] = (double)c.z;
            q[6] = (double)c.w;
        }
        __syncthreads();
        const int m = __builtin_amdgcn_readfirstlane(min(LT, n - base));
        if (live) {
            // PU points per step: their LDS reads are in flight together and the (rare) undecided ones are settled once
            // per step, one branch per PU points instead of one per point (PU = 3: 116 VGPRs, 4 wavefronts per SIMD;
            // 4 takes 184.  Two or four hypotheses per lane, to halve the broadcast reads, were slower: 3.6 / 3.8 ms
            // against 2.6 ms on the 10 000-hypothesis configuration -- the kernel is FP64-issue bound, not LDS bound.)
            constexpr int PU = 3;
            int i = 0;
            for (; i + PU <= m; i += PU) {
                int f[PU], cnt = 0;
                bool undecided = false;
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    f[u] = classify(i + u);
                    cnt += f[u] == 1 ? 1 : 0;
                    undecided |= f[u] < 0;
                }
                if (undecided) {
#pragma unroll
                    for (int u = 0; u < PU; ++u)
                        if (f[u] < 0 && exact(i + u)) ++cnt;
                }
                total += cnt;
            }
            for (; i < m; ++i) {
                const int f = classify(i);
                total += (f == 1 || (f < 0 && exact(i))) ? 1 : 0;
            }
        }
    }
    if (have) inlier_count[o] = total;
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-precision pre-classification (rectified rigs, counting only).
//
// The FP64 classification above costs ~24 half-rate instructions per point and hypothesis.  Almost every point is far
// from the 4-pixel circle, so a single-precision evaluation WITH A RIGOROUS ERROR BOUND decides it: the point counts as
// inside (outside) only if it is inside (outside) by more than everything single precision can have lost; the rest --
// about one test in a thousand on a hypothesis near the truth, far fewer on a random one -- is settled by the FP64
// functions above, so the counts stay the reference's bit for bit.  ~28 full-rate instructions per test instead of ~24 half-rate ones + 4 compares.
//
// Quantities: u = 2^-24.  Rounding x, y, z and the matrix to float and evaluating
//     h = fma(x, p0, fma(y, p1, fma(z, p2, p3)))
// loses at most 5.01 u H with H = |p0 x| + |p1 y| + |p2 z| + |p3| (2 u from the operands, u per fma); the right camera's
// hx = hxl + (pr3 - p3) at most 8 u (H + |pr3|).  a = fma(-obs, hz, hx) (the observation is a float, exact) is then off
// by at most e_hx + |obs| e_hz + u |a|, so the distance r = sqrt(a^2 + b^2) and the radius thr |hz| together by at most
//     E = 8 u (Hx + Hy) + 6 u (|obs_x| + |obs_y| + thr) Hz   + a few u of r and of the radius.
// With s >= |x| + |y| + |z| per point (pack_points_kernel) and G >= |obs_x| + |obs_y| + thr per segment,
//     Hx + Hy <= (mx + my) s + |p3| + |pr3| + |p7|,   Hz <= mz s + |p11|        (m* = largest |entry| of a row's 3x3 part)
// so E <= cs s + c0 with two constants per hypothesis; they are computed with 10 u instead of 8 u / 6 u and a further
// 0.1 %, which also covers their own rounding and the difference (< 5e-16 H) between the exact homogeneous coordinates
// and the FP64 ones the reference computes.  The relative terms (u of r, the rounding of thr |hz|, of L, U and of their
// squares: < 1e-6 together) are covered by the factors 1 -+ 2e-6 on the radius.  Decision:
//     inside  iff  q < L |L|  in both cameras,  L = thr |hz| (1 - 2e-6) - E      (L |L| < 0 when L < 0: never inside)
//     outside iff  q > U^2    in one camera,    U = thr |hz| (1 + 2e-6) + E
// (an intermediate that overflows has H > 3.4e38, hence E > 1e32 and U^2 = inf: nothing is greater, undecided; with
// hz = +-inf the left-hand sides are inf or NaN, never less than L |L|)
// A decisive answer is therefore at least 1e-7 (relative) away from the radius, where the FP64 classification (band
// 1e-9) and the reference's own arithmetic agree with exact arithmetic.  Anything irregular is undecided by
// construction: a NaN or infinity anywhere makes both comparisons false; s = NaN marks a point with a non-finite or
// denormal-range coordinate, cs = NaN a matrix with such an entry (their conversion to float is not relative-u
// accurate); c0 carries 1e-17 (+ 1e-30 G) for products that underflow.
struct HypF {
    float p[12];      // left projection matrix
    float d3;         // right matrix element (0, 3) minus the left one
    float cs, c0;     // error bound E = cs * s + c0
};

constexpr double F32_U = 5.9604644775390625e-8;        // 2^-24
constexpr float F32_CLO = 1.0f - 2e-6f, F32_CHI = 1.0f + 2e-6f;

__device__ __forceinline__ bool f32_unsafe(double v) {   // not convertible to float with relative error u
    const double a = fabs(v);
    return !(a <= 3.0e38) || (a != 0.0 && a < 1e-30);
}

// Pl: left projection matrix (FP64, as the exact path uses it), pr3: right element (0, 3), G: see above
__device__ __forceinline__ void make_hyp_f(const double *Pl, double pr3, double G, HypF &h) {
    bool bad = f32_unsafe(pr3);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        h.p[i] = (float)Pl[i];
        bad |= f32_unsafe(Pl[i]);
    }
    h.d3 = (float)(pr3 - Pl[3]);
    const double mx = fmax(fabs(Pl[0]), fmax(fabs(Pl[1]), fabs(Pl[2])));
    const double my = fmax(fabs(Pl[4]), fmax(fabs(Pl[5]), fabs(Pl[6])));
    const double mz = fmax(fabs(Pl[8]), fmax(fabs(Pl[9]), fabs(Pl[10])));
    const double k = 10.0 * F32_U * 1.001;
    const double cs = k * ((mx + my) + G * mz);
    const double c0 = k * (fabs(Pl[3]) + fabs(pr3) + fabs(Pl[7]) + G * fabs(Pl[11])) + 1e-17 + 1e-30 * G;
    h.cs = bad ? __int_as_float(0x7fc00000) : (float)cs;
    h.c0 = (float)c0;
    asm volatile("" : "+v"(h.cs));     // opaque: keeps the select above out of the per-point loop
}

// bit 0: inside both cameras for certain; bit 1: outside one camera for certain; neither: undecided
__device__ __forceinline__ int classify_f32(const HypF &h, float x, float y, float z, float s, float ul, float vl,
                                            float ur, float vr, float thr) {
    const float hz = fmaf(x, h.p[8], fmaf(y, h.p[9], fmaf(z, h.p[10], h.p[11])));
    const float hy = fmaf(x, h.p[4], fmaf(y, h.p[5], fmaf(z, h.p[6], h.p[7])));
    const float hxl = fmaf(x, h.p[0], fmaf(y, h.p[1], fmaf(z, h.p[2], h.p[3])));
    const float hxr = hxl + h.d3;
    const float al = fmaf(-ul, hz, hxl), bl = fmaf(-vl, hz, hy), ar = fmaf(-ur, hz, hxr), br = fmaf(-vr, hz, hy);
    const float ql = fmaf(al, al, bl * bl), qr = fmaf(ar, ar, br * br);
    const float A = thr * fabsf(hz);
    const float E = fmaf(s, h.cs, h.c0);
    const float L = fmaf(A, F32_CLO, -E), U = fmaf(A, F32_CHI, E);
    const float L2 = L * fabsf(L), U2 = U * U;
    const int in = (ql < L2 && qr < L2) ? 1 : 0;
    const int out = (ql > U2 || qr > U2) ? 2 : 0;
    return in | out;
}

// lane j's value in every lane, held in a VECTOR register (the empty asm pins it there).  Scalar registers would do for
// a wave-uniform constant, but 15 constants x HW hypotheses next to the by-value K and M2 exceed them, and a spilled
// scalar comes back through v_readlane at every use.
__device__ __forceinline__ float lane_to_all(float v, int j) {
    float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
    asm volatile("" : "+v"(r));
    return r;
}

// left projection matrix and the right one's element (0, 3) of a hypothesis, the operations of gemm_33_34 / compose_34
// for exactly these entries (rectified rig: every other right entry equals the left one, see within_fast_rect)
__device__ __forceinline__ void rect_matrices(const Mat33 &K, const Mat34 &M2, const double *T, double *Pl, double &pr3) {
    double Tr[12];
    compose_34(M2.m, T, Tr);
    gemm_33_34(K.m, T, Pl);
    double Pr[12];
    gemm_33_34(K.m, Tr, Pr);
    pr3 = Pr[3];
}

// The reference's answer for one point, through the FP64 classification and, inside its band, the literal arithmetic.
__device__ __forceinline__ bool settle_f64(const double *Pl, double pr3, double x, double y, double z, double ul, double vl,
                                           double ur, double vr, double thr) {
    const double lo = thr * thr * (1.0 - 1e-9), hi = thr * thr * (1.0 + 1e-9);
    const int f = within_fast_rect(Pl, pr3, x, y, z, ul, vl, ur, vr, lo, hi);
    if (f >= 0) return f == 1;
    double R[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) R[e] = Pl[e];
    R[3] = pr3;
    return within(Pl, x, y, z, ul, vl, thr) && within(R, x, y, z, ur, vr, thr);
}

// Points of every segment as 32-byte single-precision records {x, y, z, s}, {ul, vl, ur, vr} and the segment's
// observation magnitude scale[seg] >= |obs_x| + |obs_y| over its points (see above).  One workgroup per segment.
__global__ __launch_bounds__(256) void pack_points_kernel(const double *X, const float4 *xy1, const int *count, int seg_stride,
                                                          float4 *packed, float *scale) {
    __shared__ float red[4];
    const int seg = blockIdx.x;
    const int n = min(count[seg], seg_stride);
    float g = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const size_t o = (size_t)seg * seg_stride + i;
        const double x = X[o * 3 + 0], y = X[o * 3 + 1], z = X[o * 3 + 2];
        const float4 c = xy1[o];
        const double s = (fabs(x) + fabs(y) + fabs(z)) * (1.0 + 1e-6);
        const float gi = (fmaxf(fabsf(c.x), fabsf(c.z)) + fmaxf(fabsf(c.y), fabsf(c.w))) * (1.0f + 1e-6f);
        const bool bad = f32_unsafe(x) || f32_unsafe(y) || f32_unsafe(z) || !(s <= 3.0e38) || !(gi <= 3.0e38f);
        if (!bad) g = fmaxf(g, gi);
        packed[o * 2 + 0] = make_float4((float)x, (float)y, (float)z, bad ? __int_as_float(0x7fc00000) : (float)s);
        packed[o * 2 + 1] = c;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) g = fmaxf(g, __shfl_xor(g, m));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = g;
    __syncthreads();
    if (threadIdx.x == 0) scale[seg] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// score_kernel with the single-precision pre-classification: one wavefront scores HW hypotheses of a segment, a lane
// takes a point.
constexpr int QCAP = 64;     // undecided tests a wavefront can queue (one per lane of the settling pass)

template <int HW>
__global__ __launch_bounds__(SB) void score_f32_kernel(Mat33 K, Mat34 M2, const double *X, const float4 *xy1,
                                                       const float4 *packed, const float *scale, const int *count,
                                                       const int *seg_of, int n_rows, int groups, int seg_stride,
                                                       const double *pose, const uint8_t *status, int n_hyp, double thr,
                                                       int *inlier_count) {
    const int xcd = blockIdx.x % 8, k = blockIdx.x / 8;          // a segment's workgroups on one XCD, as in score_kernel
    const int row = (k / groups) * 8 + xcd;
    if (row >= n_rows) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h0 = ((k % groups) * (SB / 64) + wave) * HW;
    if (h0 >= n_hyp) return;
    const int seg = seg_of ? seg_of[row] : row;
    const int n = min(count[seg], seg_stride);
    const double G = (double)scale[seg] + thr;
    const float thrf = (float)thr;

    // Set-up: lane j builds hypothesis h0 + j (the FP64 matrices, then the single-precision constants), so the HW
    // set-ups cost one; v_readlane then hands every lane all HW sets of constants (lane_to_all).
    HypF H[HW];
    bool live[HW];
    bool any_live = false;
    {
        const int jl = lane < HW ? lane : HW - 1;
        const int h = h0 + jl < n_hyp ? h0 + jl : h0;
        const size_t o = (size_t)row * n_hyp + h;
        double T[12], Pl[12], pr3;
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = pose[o * 12 + i];
        rect_matrices(K, M2, T, Pl, pr3);
        bool ok = h0 + jl < n_hyp && status[o] != 0;
        // a NaN entry makes a homogeneous coordinate NaN for every point, hence the comparison false: no supporters
#pragma unroll
        for (int i = 0; i < 12; ++i) ok = ok && !(Pl[i] != Pl[i]);
        ok = ok && !(pr3 != pr3);
        HypF hf;
        make_hyp_f(Pl, pr3, G, hf);
        const unsigned long long okm = __ballot(ok);
#pragma unroll
        for (int j = 0; j < HW; ++j) {
#pragma unroll
            for (int i = 0; i < 12; ++i) H[j].p[i] = lane_to_all(hf.p[i], j);
            H[j].d3 = lane_to_all(hf.d3, j);
            H[j].cs = lane_to_all(hf.cs, j);
            H[j].c0 = lane_to_all(hf.c0, j);
            live[j] = (okm >> j) & 1ull;
            any_live |= live[j];
        }
    }
    int total[HW];
#pragma unroll
    for (int j = 0; j < HW; ++j) total[j] = 0;
    const float4 *pk = packed + (size_t)seg * seg_stride * 2;
    // Undecided tests (a handful per wavefront: HW x n tests at one in a thousand or fewer) are queued in LDS, per
    // wavefront, and settled in FP64 after the scan, so that the scan itself holds nothing but the single-precision
    // constants: (point << 3 | hypothesis slot).  Overflow and irregular values (an infinite intermediate has H > 3e38,
    // hence E > 1e32 and U^2 = inf: undecided) are covered: see the recount below and the header.
    __shared__ int q_cnt[SB / 64];
    __shared__ int q_ent[SB / 64][QCAP];
    if (lane == 0) q_cnt[wave] = 0;
    if (any_live) {
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            const bool have = i < n;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (have) {
                a = pk[(size_t)i * 2];
                b = pk[(size_t)i * 2 + 1];
            }
#pragma unroll
            for (int j = 0; j < HW; ++j) {
                if (!live[j]) continue;                       // wave-uniform
                const int f = classify_f32(H[j], a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, thrf);
                total[j] += __popcll(__builtin_amdgcn_ballot_w64(have && (f & 1)));
                if (have && f == 0) {
                    const int slot = atomicAdd(&q_cnt[wave], 1);
                    if (slot < QCAP) q_ent[wave][slot] = (i << 3) | j;
                }
            }
        }
        // LDS operations of one wavefront complete in order: the count read here includes every queued entry
        const int queued = __builtin_amdgcn_readfirstlane(q_cnt[wave]);
        const double *Xs = X + (size_t)seg * seg_stride * 3;
        const float4 *obs = xy1 + (size_t)seg * seg_stride;
        if (queued > 0 && queued <= QCAP) {
            // lane e settles entry e on its hypothesis' FP64 matrices, rebuilt here
            bool in = false;
            int ej = -1;
            if (lane < queued) {
                const int ent = q_ent[wave][lane];
                const int i = ent >> 3;
                ej = ent & 7;
                const size_t o = (size_t)row * n_hyp + h0 + ej;
                double T[12], Pl[12], pr3;
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] = pose[o * 12 + e];
                rect_matrices(K, M2, T, Pl, pr3);
                const float4 c = obs[i];
                in = settle_f64(Pl, pr3, Xs[i * 3 + 0], Xs[i * 3 + 1], Xs[i * 3 + 2], (double)c.x, (double)c.y, (double)c.z,
                                (double)c.w, thr);
            }
#pragma unroll
            for (int j = 0; j < HW; ++j) total[j] += __popcll(__builtin_amdgcn_ballot_w64(in && ej == j));
        } else if (queued > QCAP) {
            // more undecided tests than the queue holds (nothing on real data does this): count again in FP64
#pragma unroll 1
            for (int j = 0; j < HW; ++j) {
                if (!live[j]) continue;
                const size_t o = (size_t)row * n_hyp + h0 + j;
                double T[12], Pl[12], pr3;
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] = pose[o * 12 + e];
                rect_matrices(K, M2, T, Pl, pr3);
                int t = 0;
                for (int base = 0; base < n; base += 64) {
                    const int i = base + lane;
                    bool in = false;
                    if (i < n) {
                        const float4 c = obs[i];
                        in = settle_f64(Pl, pr3, Xs[i * 3 + 0], Xs[i * 3 + 1], Xs[i * 3 + 2], (double)c.x, (double)c.y,
                                        (double)c.z, (double)c.w, thr);
                    }
                    t += __popcll(__builtin_amdgcn_ballot_w64(in));
                }
                total[j] = t;
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < HW; ++j)
            if (h0 + j < n_hyp) inlier_count[(size_t)row * n_hyp + h0 + j] = total[j];
    }
}

// score_lanes_kernel with the single-precision pre-classification: HL hypotheses per lane, the segment's records staged
// through LDS in tiles of LTF points and read back as broadcasts (two ds_read_b128 per point feed 64 x HL tests; one
// hypothesis per lane measured faster than two: 1.34 against 1.20 T tests/s, occupancy and grid fill).
// Undecided tests are not rare enough to be an afterthought: a hypothesis near the truth puts many points a few pixels
// from their observations, and about one of its tests in a thousand falls inside the single-precision band (+-1e-3 px).
// They are queued per wavefront in LDS as (lane, hypothesis slot, point) and settled COOPERATIVELY: each lane takes one
// queued test, whoever queued it -- rebuilds that hypothesis' FP64 matrices and runs the FP64 functions -- and credits
// the owner's counter in LDS.  All lanes work whatever the distribution of the tests over the lanes; the queue is
// drained whenever the next step could overflow it, so nothing is ever dropped or recounted.
// (The records are wave-uniform, so the scalar data path can deliver them too -- no LDS, no barrier: measured, 13.7 ms
// against 15.7 for the loop-closure configuration's 570-point segments, but 5.9 ms against 3.3 on 6000-point segments,
// whose 190 KB of records stream through the 16 KB scalar cache; scalar loads return out of order, so a wavefront
// cannot keep more than one batch in flight.)
constexpr int LTF = 256;      // points per LDS tile (32 bytes each: 8 KB; with the queue 17 KB per workgroup)
#ifndef SLAM_F32_HL
#define SLAM_F32_HL 1
#endif
constexpr int HL = SLAM_F32_HL;   // hypotheses per lane
#ifndef SLAM_F32_PU
#define SLAM_F32_PU 4
#endif
constexpr int PU = SLAM_F32_PU;   // points per step of the scan
// registers capped for five wavefronts per SIMD (the few spills land in the set-up and the drain): 1.33 -> 1.40 T tests/s;
// six spill inside the drain (0.39 where drains are frequent)
#ifndef SLAM_F32_WPE
#define SLAM_F32_WPE 5
#endif
constexpr int WQCAP = 512;        // queued tests per wavefront; a step adds at most 64 * PU * HL
static_assert(64 * PU * HL <= WQCAP / 2, "a step must fit the queue's reserve");

__global__ __launch_bounds__(SB) __attribute__((amdgpu_waves_per_eu(SLAM_F32_WPE, 8))) void score_lanes_f32_kernel(Mat33 K, Mat34 M2, const double *X, const float4 *xy1,
                                                             const float4 *__restrict__ packed, const float *scale,
                                                             const int *count, const int *seg_of, int n_rows, int groups,
                                                             int seg_stride, const double *pose, const uint8_t *status,
                                                             int n_hyp, double thr, int *__restrict__ inlier_count) {
    __shared__ float4 tile[LTF * 2];
    __shared__ unsigned wq[SB / 64][WQCAP];
    __shared__ int wq_n[SB / 64];
    __shared__ int credit[SB / 64][HL * 64];
    const int xcd = blockIdx.x % 8, k = blockIdx.x / 8;
    const int row = (k / groups) * 8 + xcd;
    if (row >= n_rows) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane(seg_of ? seg_of[row] : row);
    const int n = __builtin_amdgcn_readfirstlane(min(count[seg], seg_stride));
    const double G = (double)scale[seg] + thr;
    const float thrf = (float)thr;
    const int h_first = (k % groups) * HL * SB;          // slot e of thread t: hypothesis h_first + e * SB + t

    HypF H[HL];
    bool live[HL];
    bool any_live = false;
#pragma unroll
    for (int e = 0; e < HL; ++e) {
        const int h = h_first + e * SB + threadIdx.x;                 // consecutive lanes, consecutive hypotheses
        const bool have = h < n_hyp;
        const size_t o = (size_t)row * n_hyp + (have ? h : 0);
        double T[12], Pl[12], pr3;
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = pose[o * 12 + i];
        rect_matrices(K, M2, T, Pl, pr3);
        bool ok = have && status[o] != 0;
        // a NaN entry makes a homogeneous coordinate NaN for every point, hence the comparison false: no supporters
#pragma unroll
        for (int i = 0; i < 12; ++i) ok = ok && !(Pl[i] != Pl[i]);
        ok = ok && !(pr3 != pr3);
        live[e] = ok;
        any_live |= ok;
        make_hyp_f(Pl, pr3, G, H[e]);
        credit[wave][e * 64 + lane] = 0;
    }
    if (lane == 0) wq_n[wave] = 0;
    const bool wave_live = __ballot(any_live) != 0ull;
    const float4 *pk = packed + (size_t)seg * seg_stride * 2;

    // settles the wavefront's queued tests, one per lane and round (LDS operations of a wavefront complete in order)
    auto drain = [&]() {
        __builtin_amdgcn_wave_barrier();
        const int queued = __builtin_amdgcn_readfirstlane(wq_n[wave]);
        const double *Xs = X + (size_t)seg * seg_stride * 3;
        const float4 *obs = xy1 + (size_t)seg * seg_stride;
        for (int r = 0; r < queued; r += 64) {
            const bool valid = r + lane < queued;
            const unsigned enc = valid ? wq[wave][r + lane] : 0u;
            const int g = enc & 0xffff, e = (enc >> 16) & (HL - 1), owner = enc >> 24;
            const size_t o = (size_t)row * n_hyp + h_first + e * SB + wave * 64 + owner;
            if (valid) {
                double T[12], Pl[12], pr3;
#pragma unroll
                for (int c = 0; c < 12; ++c) T[c] = pose[o * 12 + c];
                rect_matrices(K, M2, T, Pl, pr3);
                const float4 c = obs[g];
                if (settle_f64(Pl, pr3, Xs[(size_t)g * 3 + 0], Xs[(size_t)g * 3 + 1], Xs[(size_t)g * 3 + 2], (double)c.x, (double)c.y,
                               (double)c.z, (double)c.w, thr))
                    atomicAdd(&credit[wave][e * 64 + owner], 1);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) wq_n[wave] = 0;
        __builtin_amdgcn_wave_barrier();
    };

    int total[HL];
#pragma unroll
    for (int e = 0; e < HL; ++e) total[e] = 0;
    for (int base = 0; base < n; base += LTF) {
        if (base) __syncthreads();                      // the previous tile has been consumed
        const int m = min(LTF, n - base);
        for (int t = threadIdx.x; t < 2 * m; t += SB) tile[t] = pk[(size_t)base * 2 + t];
        __syncthreads();
        if (!wave_live) continue;
        // PU points per step: their LDS reads are in flight together, the PU x HL tests form one block of independent
        // instruction chains, and the undecided ones are queued once per step, one branch instead of PU x HL
        auto step = [&](int i, auto pu_tag) -> bool {
            constexpr int PU_ = decltype(pu_tag)::value;
            float4 a[PU_], b[PU_];
#pragma unroll
            for (int u = 0; u < PU_; ++u) {
                a[u] = tile[2 * (i + u)];
                b[u] = tile[2 * (i + u) + 1];
            }
            // which lanes are undecided, per test, as wave masks in scalar registers (no per-lane bookkeeping in the scan)
            unsigned long long und[PU_ * HL], any = 0ull;
#pragma unroll
            for (int u = 0; u < PU_; ++u)
#pragma unroll
                for (int e = 0; e < HL; ++e) {
                    const int f = classify_f32(H[e], a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, b[u].z, b[u].w, thrf);
                    total[e] += f & 1;
                    und[u * HL + e] = __builtin_amdgcn_ballot_w64(f == 0 && live[e]);
                    any |= und[u * HL + e];
                }
            if (any) {
#pragma unroll
                for (int t = 0; t < PU_ * HL; ++t)
                    if (und[t] >> lane & 1ull) {
                        const int slot = atomicAdd(&wq_n[wave], 1);          // < WQCAP: drained below before it can fill
                        wq[wave][slot] = (unsigned)lane << 24 | (unsigned)(t % HL) << 16 | (unsigned)(base + i + t / HL);
                    }
                __builtin_amdgcn_wave_barrier();
                return true;
            }
            return false;
        };
        // the queue is drained between steps (no point record is live then) once it is half full
        int i = 0;
        for (; i + PU <= m; i += PU)
            if (step(i, std::integral_constant<int, PU>()) && __builtin_amdgcn_readfirstlane(wq_n[wave]) > WQCAP / 2) drain();
        for (; i < m; ++i)
            if (step(i, std::integral_constant<int, 1>()) && __builtin_amdgcn_readfirstlane(wq_n[wave]) > WQCAP / 2) drain();
    }
    if (wave_live) drain();
#pragma unroll
    for (int e = 0; e < HL; ++e) {
        const int h = h_first + e * SB + threadIdx.x;
        if (h < n_hyp) inlier_count[(size_t)row * n_hyp + h] = live[e] ? total[e] + credit[wave][e * 64 + lane] : 0;
    }
}

// drop-in consensus_match: float64 observations, one pose, byte mask
__global__ __launch_bounds__(256) void consensus_f64_kernel(Mat33 K, Mat34 M2, Mat34 T, const double *X,
                                                            const double2 *l1, const double2 *r1, int n, double thr,
                                                            uint8_t *mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double Tr[12], Pl[12], Pr[12];
    compose_34(M2.m, T.m, Tr);
    gemm_33_34(K.m, T.m, Pl);
    gemm_33_34(K.m, Tr, Pr);
    const double x = X[(size_t)i * 3], y = X[(size_t)i * 3 + 1], z = X[(size_t)i * 3 + 2];
    const double2 a = l1[i], b = r1[i];
    mask[i] = within(Pl, x, y, z, a.x, a.y, thr) && within(Pr, x, y, z, b.x, b.y, thr);
}

// K = [fx 0 cx; 0 fy cy; 0 0 1] and M2 = [I | (tx, 0, 0)], element for element (the KITTI rig of the reference; any other
// rig takes the general kernel)
bool rectified_rig(const Mat33 &K, const Mat34 &M2) {
    const double *k = K.m, *m = M2.m;
    return k[1] == 0.0 && k[3] == 0.0 && k[6] == 0.0 && k[7] == 0.0 && k[8] == 1.0 && m[0] == 1.0 && m[1] == 0.0 &&
           m[2] == 0.0 && m[4] == 0.0 && m[5] == 1.0 && m[6] == 0.0 && m[7] == 0.0 && m[8] == 0.0 && m[9] == 0.0 &&
           m[10] == 1.0 && m[11] == 0.0 && std::isfinite(k[0]) && std::isfinite(k[2]) && std::isfinite(k[4]) &&
           std::isfinite(k[5]) && std::isfinite(m[3]);
}

}  // namespace

int slam_launch_score(const Mat33 &K, const Mat34 &M2, const double *d_X, const void *d_xy1, int xy_f64,
                      const int32_t *d_count,
                      const int32_t *d_seg_of, int n_rows, int seg_stride, const double *d_pose,
                      const uint8_t *d_status, int n_hyp, double thr, int32_t *d_inlier_count, uint64_t *d_mask,
                      hipStream_t st, const int32_t *d_gate, const float *d_packed, const float *d_scale) {
    if (n_rows <= 0 || n_hyp <= 0) return SLAM_OK;
    SLAM_REQUIRE(!d_gate || d_mask, "slam_score_hypotheses: a gate is supported on the mask path only");
    SLAM_REQUIRE(!d_mask || seg_stride % 64 == 0, "slam_score_hypotheses: seg_stride must be a multiple of 64 for masks");
    const bool rect = rectified_rig(K, M2);
    // counting on a rectified rig with the points' single-precision records at hand (slam_launch_pack_points): the
    // pre-classified kernels; same counts.  A radius outside [1e-3, 1e6] is not covered by their error analysis.
    if (!d_mask && rect && d_packed && d_scale && !xy_f64 && thr >= 1e-3 && thr <= 1e6 &&
        (n_hyp < LANES_MIN_HYP || seg_stride <= 65536)) {      // the lane kernel queues 16-bit point indices
        const bool lanes = n_hyp >= LANES_MIN_HYP;
        const int hw = n_hyp >= 16 ? 4 : 2;
        const int groups = lanes ? slam_div_up(n_hyp, SB * HL) : slam_div_up(n_hyp, (SB / 64) * hw);
        const long n_blocks = (long)groups * (slam_div_up(n_rows, 8) * 8L);
        SLAM_REQUIRE(n_blocks < (1L << 31), "slam_score_hypotheses: %d segments x %d hypotheses is too large for one call", n_rows, n_hyp);
        const dim3 grid((unsigned)n_blocks), block(SB);
#define SLAM_F32_ARGS K, M2, d_X, (const float4 *)d_xy1, (const float4 *)d_packed, d_scale, d_count, d_seg_of, n_rows, groups, \
                      seg_stride, d_pose, d_status, n_hyp, thr, d_inlier_count
        if (lanes) hipLaunchKernelGGL(score_lanes_f32_kernel, grid, block, 0, st, SLAM_F32_ARGS);
        else if (hw == 4) hipLaunchKernelGGL((score_f32_kernel<4>), grid, block, 0, st, SLAM_F32_ARGS);
        else hipLaunchKernelGGL((score_f32_kernel<2>), grid, block, 0, st, SLAM_F32_ARGS);
#undef SLAM_F32_ARGS
        SLAM_HIP_CHECK(hipGetLastError());
        return SLAM_OK;
    }
    if (!d_mask && n_hyp >= LANES_MIN_HYP) {
        // counting only and enough hypotheses per segment to fill workgroups with them: one lane per hypothesis
        const int groups = slam_div_up(n_hyp, SB);
        const long n_blocks = (long)groups * (slam_div_up(n_rows, 8) * 8L);
        SLAM_REQUIRE(n_blocks < (1L << 31), "slam_score_hypotheses: %d segments x %d hypotheses is too large for one call", n_rows, n_hyp);
        const dim3 grid((unsigned)n_blocks), block(SB);
#define SLAM_LANES_LAUNCH(XY, RECT)                                                                                      \
    hipLaunchKernelGGL((score_lanes_kernel<XY, RECT>), grid, block, 0, st, K, M2, d_X, (const XY *)d_xy1, d_count, d_seg_of, \
                       n_rows, groups, seg_stride, d_pose, d_status, n_hyp, thr, d_inlier_count)
        if (xy_f64) {
            if (rect) SLAM_LANES_LAUNCH(double4, true); else SLAM_LANES_LAUNCH(double4, false);
        } else {
            if (rect) SLAM_LANES_LAUNCH(float4, true); else SLAM_LANES_LAUNCH(float4, false);
        }
#undef SLAM_LANES_LAUNCH
        SLAM_HIP_CHECK(hipGetLastError());
        return SLAM_OK;
    }
    // counting only: two hypotheses per wavefront; with a mask: one (a mask row per hypothesis)
    const int hw = d_mask ? 1 : 2;
    const int groups = slam_div_up(n_hyp, (SB / 64) * hw);                 // workgroups per segment
    const long n_blocks = (long)groups * (slam_div_up(n_rows, 8) * 8L);
    SLAM_REQUIRE(n_blocks < (1L << 31), "slam_score_hypotheses: %d segments x %d hypotheses is too large for one call", n_rows, n_hyp);
    const dim3 grid((unsigned)n_blocks), block(SB);
#define SLAM_SCORE_LAUNCH(XY, HW, RECT)                                                                                    \
    hipLaunchKernelGGL((score_kernel<XY, HW, RECT>), grid, block, 0, st, K, M2, d_X, (const XY *)d_xy1, d_count, d_seg_of, \
                       n_rows, groups, seg_stride, d_pose, d_status, n_hyp, thr, d_inlier_count, (unsigned long long *)d_mask, d_gate)
#define SLAM_SCORE_LAUNCH_HW(XY, RECT)                                          \
    do {                                                                        \
        if (hw == 2) SLAM_SCORE_LAUNCH(XY, 2, RECT); else SLAM_SCORE_LAUNCH(XY, 1, RECT); \
    } while (0)
    if (xy_f64) {
        if (rect) SLAM_SCORE_LAUNCH_HW(double4, true); else SLAM_SCORE_LAUNCH_HW(double4, false);
    } else {
        if (rect) SLAM_SCORE_LAUNCH_HW(float4, true); else SLAM_SCORE_LAUNCH_HW(float4, false);
    }
#undef SLAM_SCORE_LAUNCH_HW
#undef SLAM_SCORE_LAUNCH
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

int slam_launch_pack_points(const double *d_X, const float *d_xy1, const int32_t *d_count, int n_seg, int seg_stride,
                            float *d_packed, float *d_scale, hipStream_t st) {
    if (n_seg <= 0 || seg_stride <= 0) return SLAM_OK;
    SLAM_REQUIRE(d_X && d_xy1 && d_count && d_packed && d_scale, "slam_score_pack_points: null pointer");
    SLAM_REQUIRE((((uintptr_t)d_packed | (uintptr_t)d_xy1) & 15) == 0, "slam_score_pack_points: d_xy1 and d_packed must be 16-byte aligned");
    hipLaunchKernelGGL(pack_points_kernel, dim3(n_seg), dim3(256), 0, st, d_X, (const float4 *)d_xy1, d_count, seg_stride,
                       (float4 *)d_packed, d_scale);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_score_pack_points(const double *d_X, const float *d_xy1, const int32_t *d_count, int n_seg,
                                      int seg_stride, float *d_packed, float *d_scale, void *stream) {
    return slam_launch_pack_points(d_X, d_xy1, d_count, n_seg, seg_stride, d_packed, d_scale, (hipStream_t)stream);
}

extern "C" int slam_score_hypotheses_packed(const double *h_K, const double *h_M2, const double *d_X, const float *d_xy1,
                                            const float *d_packed, const float *d_scale, const int32_t *d_count, int n_seg,
                                            int seg_stride, const double *d_pose, const uint8_t *d_status, int n_hyp,
                                            double thr, int32_t *d_inlier_count, void *stream) {
    SLAM_REQUIRE(h_K && h_M2 && d_X && d_xy1 && d_packed && d_scale && d_count && d_pose && d_status && d_inlier_count,
                 "slam_score_hypotheses_packed: null pointer");
    SLAM_REQUIRE((((uintptr_t)d_packed | (uintptr_t)d_xy1) & 15) == 0, "slam_score_hypotheses_packed: d_xy1 and d_packed must be 16-byte aligned");
    SLAM_REQUIRE(n_seg >= 0 && seg_stride >= 0 && n_hyp >= 0, "slam_score_hypotheses_packed: negative size");
    Mat33 K;
    Mat34 M2;
    for (int i = 0; i < 9; ++i) K.m[i] = h_K[i];
    for (int i = 0; i < 12; ++i) M2.m[i] = h_M2[i];
    return slam_launch_score(K, M2, d_X, d_xy1, 0, d_count, nullptr, n_seg, seg_stride, d_pose, d_status, n_hyp, thr,
                             d_inlier_count, nullptr, (hipStream_t)stream, nullptr, d_packed, d_scale);
}

extern "C" int slam_score_hypotheses(const double *h_K, const double *h_M2, const double *d_X, const void *d_xy1,
                                     int xy_f64, const int32_t *d_count, int n_seg, int seg_stride, const double *d_pose,
                                     const uint8_t *d_status, int n_hyp, double thr, int32_t *d_inlier_count,
                                     uint64_t *d_mask, void *stream) {
    SLAM_REQUIRE(h_K && h_M2 && d_X && d_xy1 && d_count && d_pose && d_status && d_inlier_count,
                 "slam_score_hypotheses: null pointer");
    Mat33 K;
    Mat34 M2;
    for (int i = 0; i < 9; ++i) K.m[i] = h_K[i];
    for (int i = 0; i < 12; ++i) M2.m[i] = h_M2[i];
    return slam_launch_score(K, M2, d_X, d_xy1, xy_f64, d_count, nullptr, n_seg, seg_stride, d_pose, d_status, n_hyp, thr,
                             d_inlier_count, d_mask, (hipStream_t)stream);
}

extern "C" int slam_consensus_match_f64(const double *h_K, const double *h_M2, const double *h_pose,
                                        const double *d_X, const double *d_left1, const double *d_right1, int n,
                                        double thr, uint8_t *d_mask, void *stream) {
    if (n <= 0) return SLAM_OK;
    SLAM_REQUIRE(h_K && h_M2 && h_pose && d_X && d_left1 && d_right1 && d_mask, "slam_consensus_match_f64: null pointer");
    Mat33 K;
    Mat34 M2, T;
    for (int i = 0; i < 9; ++i) K.m[i] = h_K[i];
    for (int i = 0; i < 12; ++i) {
        M2.m[i] = h_M2[i];
        T.m[i] = h_pose[i];
    }
    hipLaunchKernelGGL(consensus_f64_kernel, dim3(slam_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, K, M2, T,
                       d_X, (const double2 *)d_left1, (const double2 *)d_right1, n, thr, d_mask);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

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
                      hipStream_t st, const int32_t *d_gate) {
    if (n_rows <= 0 || n_hyp <= 0) return SLAM_OK;
    SLAM_REQUIRE(!d_gate || d_mask, "slam_score_hypotheses: a gate is supported on the mask path only");
    SLAM_REQUIRE(!d_mask || seg_stride % 64 == 0, "slam_score_hypotheses: seg_stride must be a multiple of 64 for masks");
    const bool rect = rectified_rig(K, M2);
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

// PnP refit on the supporters: Levenberg-Marquardt on the left-image reprojection error, one workgroup of four
// wavefronts per frame pair.
//
// Replaces extra_refine -> pnp(world[mask], left1[mask], K, cv2.SOLVEPNP_ITERATIVE) (utils/utills.py:548-571,
// :585-596).  cv2's ITERATIVE solver is a DLT initialisation followed by Levenberg-Marquardt (CvLevMarq, at
// most 20 iterations, relative step < FLT_EPSILON) on the pixel reprojection error of the LEFT camera only;
// it therefore returns the local least-squares minimum to ~1e-7 relative.  This kernel minimises the same
// cost to convergence, started from the winning RANSAC hypothesis (which already explains every supporter to
// within 4 px, so it lies in the same basin), with a left-multiplied se(3) increment instead of OpenCV's
// (rvec, tvec) parameters.  Same minimiser, hence the 1e-5 relative pose tolerance of the parity tests;
// OpenCV itself is absent, so this is "parity unpinned" against cv2 and pinned against oracle/slam_oracle.py.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "jacobi.h"   // rcp_nr
#include "kernels.h"

// The library is compiled with -ffp-contract=off because the matcher, triangulation and scoring reproduce numpy's
// operation order.  Nothing of that kind applies to this solver (its result is the minimiser, to which it converges
// to rounding whatever the rounding of an intermediate), so multiply-adds are fused here: about 40 % fewer FP64
// instructions in the two passes over the points.
#pragma clang fp contract(fast)

namespace {

// Sum over the 64 lanes, every lane ending with the same bits: a butterfly from stride 1 up to 32, with the partner's
// value fetched by DPP inside a row of 16 lanes and by the gfx950 row / half swaps across rows -- VALU moves, no LDS.
// (__shfl_xor compiles to two ds_bpermute_b32 per double and step; a pair's solve makes ~35 reductions of up to 27
// values, and with 12 wavefronts per CU doing that at once the LDS crossbar, not the FP64 pipe, set the pace:
// refit_kernel 0.468 -> see DESIGN section 4.)  At every step the lanes of a group already agree and two groups
// exchange their values, so both end with the same (commutative) sum whichever lane of the other group is read: quad
// swap, half-row mirror and row mirror serve as strides 4 and 8.  The tree differs from round 2's (which ran from
// stride 32 down), so poses differ from that build's in the last digits (~1e-11 relative), not between launches.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)u, CTRL, 0xF, 0xF, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(u >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);       // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_f64<0x4E>(v);       // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_f64<0x141>(v);      // row_half_mirror: the other quad of the 8
    v += dpp_f64<0x140>(v);      // row_mirror: the other half of the row
    {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);          // rows 0<->1, 2<->3
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
        v = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) +
            __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
    }
    {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);          // lanes 0-31 <-> 32-63
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
        v = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) +
            __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
    }
    return v;
}

// Cholesky solve of the 6x6 SPD system A x = b (upper triangle of A given, row-major full storage)
__device__ __forceinline__ bool chol_solve6(double (&A)[6][6], const double (&b)[6], double (&x)[6]) {
    double L[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < j) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) return false;
        const double s = sqrt(d);
        L[j][j] = s;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i > j) {
                double v = A[j][i];
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k < j) v -= L[i][k] * L[j][k];
                L[i][j] = v / s;
            }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < i) v -= L[i][k] * y[k];
        y[i] = v / L[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k > i) v -= L[k][i] * x[k];
        x[i] = v / L[i][i];
    }
    return true;
}

struct Pose { double R[3][3]; double t[3]; };

// P' = exp(d) * P with d = (omega, upsilon): R' = E R, t' = E t + upsilon
__device__ __forceinline__ Pose apply_step(const Pose &P, const double (&d)[6]) {
    const double wx = d[0], wy = d[1], wz = d[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double a, b;   // E = I + a [w]x + b [w]x^2
    if (th < 1e-8) {
        a = 1.0 - th2 / 6.0;
        b = 0.5 - th2 / 24.0;
    } else {
        a = sin(th) / th;
        b = (1.0 - cos(th)) / th2;
    }
    const double W[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    double E[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double w2 = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
            E[i][j] = (i == j ? 1.0 : 0.0) + a * W[i][j] + b * w2;
        }
    Pose Q;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Q.R[i][j] = E[i][0] * P.R[0][j] + E[i][1] * P.R[1][j] + E[i][2] * P.R[2][j];
        Q.t[i] = E[i][0] * P.t[0] + E[i][1] * P.t[1] + E[i][2] * P.t[2] + d[3 + i];
    }
    return Q;
}

constexpr int MAX_EVAL = 60;
// Stopping rule: the solve ends with the first trial step whose relative size is below FLT_EPSILON -- the size at which
// CvLevMarq, cv2.solvePnP(ITERATIVE)'s solver and so the reference's refit (utils/utills.py:562-563), ends after an
// ACCEPTED step -- whether that step is accepted (it is applied) or rejected (the pose stays; CvLevMarq itself goes on
// raising its damping there, to the same pose).  SLAM_REFIT_STEP2 is the square of that relative size.  The rule only
// applies while the damping is at most 1 (the Marquardt-scaled step is then within a factor 2 of the Gauss-Newton
// step, so a tiny step means a tiny distance to the minimum, not a heavily damped one); under heavier damping, i.e.
// after three or more rejections in a row, the solve goes on as before.  The iteration converges quadratically on top
// of a damping that is down to 1e-3 x 0.1^k by step k, so the pose after an accepted step of relative size s is within
// ~max(1e-5 s, s^2) of the minimiser: ~1e-12 here.
// Measured on bench-shaped pairs (tools/refit_evals_hist.py, a probe build counting trial steps): rounds 1-3 stopped on
// an ACCEPTED step below 1e-9 only and went on, after a rejected one, until the step was below 1e-15 relative -- at the
// minimum the cost comparison is rounding noise, so half the pairs took 5 to 16 trial steps (mean 6.3) for a pose that
// moved by < 1.2e-10; with this rule every pair takes 3 or 4 (mean 3.2), the poses differ from the fully converged ones
// by 4e-12 (median) / 2e-9 (worst) relative -- the parity bar is 1e-5 -- and the refit of a whole shard takes 0.27
// instead of 0.40 ms, that of a rank's share at N = 8 (432 pairs: the launch lasts as long as its slowest pair) 0.06
// instead of 0.10 ms.
#ifndef SLAM_REFIT_STEP2
#define SLAM_REFIT_STEP2 1.42e-14        /* FLT_EPSILON^2 */
#endif
#ifndef SLAM_REFIT_REJECT_STEP2
#define SLAM_REFIT_REJECT_STEP2 SLAM_REFIT_STEP2
#endif

// Wavefronts per pair (RW).  A pair's solve is one long dependent chain (passes over the supporters, 27 butterfly
// reductions, a serial 6x6 Cholesky, the pose update): ~100 us with one wavefront whatever the launch.  The passes over
// the points are split over RW wavefronts, the scalar part (reduced sums -> step -> new pose) being computed
// redundantly by every thread from the same bits, so control flow stays uniform over the workgroup.  RW is a
// compile-time constant of the library (SLAM_REFIT_WAVES below), never a function of the launch.
//
// LDS: the supporters of the pair, compacted once (x, y, z as doubles + the left observation in its own type: 32 bytes per
// supporter for float32 key-point rows), when there are at most `cap` of them; every Levenberg-Marquardt pass then reads them from LDS with all
// lanes busy.  (Round 1 re-read the tracked points from global memory in every pass, masked: ~15 passes x 48 KB per
// pair = 1.78 GB of fabric reads per launch for 0.2 GB of data.)  A pair with more supporters than `cap` walks global
// memory as before.
template <typename XY4, int RW>
__global__ __launch_bounds__(RW * 64) void refit_kernel(Mat33 K, const double *X, const XY4 *xy1, const int *count,
                                                        int seg_stride, const unsigned long long *mask_words,
                                                        const double *pose0, double *pose_out, uint8_t *status, int cap,
                                                        const int *gate) {
    if (gate && *gate != 0) return;          // queued speculatively behind a device-side decision that went the other way
    constexpr int RT = RW * 64;
    extern __shared__ double lds[];
    __shared__ double red[RW][28];           // per-wavefront partial sums of one reduction round
    __shared__ int wave_supp[RW];
    using UV = decltype(XY4::x);             // observations keep the caller's type: float32 key-point rows or float64
    double *sx = lds, *sy = lds + cap, *sz = lds + 2 * cap;
    UV *su = (UV *)(lds + 3 * cap), *sv = su + cap;
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = count[seg];
    const int words = seg_stride / 64;
    const int n_words = (n + 63) / 64;
    const unsigned long long *mw = mask_words + (size_t)seg * words;
    const double *Xs = X + (size_t)seg * seg_stride * 3;
    const XY4 *obs = xy1 + (size_t)seg * seg_stride;
    const double fx = K.m[0], fy = K.m[4], cx = K.m[2], cy = K.m[5];
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);

    // the 64-point mask words are dealt round-robin to the wavefronts: word j belongs to wavefront j % RW
    int mine = 0;
    for (int j = wave + RW * lane; j < n_words; j += RW * 64) {
        unsigned long long m = mw[j];
        if (j == n_words - 1 && (n & 63)) m &= (1ull << (n & 63)) - 1ull;
        mine += __popcll(m);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m);
    int supp = mine, my_offset = 0;
    if (RW > 1) {
        if (lane == 0) wave_supp[wave] = mine;
        __syncthreads();
        supp = 0;
#pragma unroll
        for (int w = 0; w < RW; ++w) {
            if (w < wave) my_offset += wave_supp[w];
            supp += wave_supp[w];
        }
    }
    if (supp < 6) {   // utills.py:558-559: pose None
        if (tid < 12) pose_out[(size_t)seg * 12 + tid] = qnan;
        if (tid == 0) status[seg] = 0;
        return;
    }

    Pose P;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) P.R[i][j] = pose0[(size_t)seg * 12 + i * 4 + j];
        P.t[i] = pose0[(size_t)seg * 12 + i * 4 + 3];
    }

    constexpr int UN = 4;
    const bool in_lds = supp <= cap;          // uniform over the workgroup
    if (in_lds) {
        // ordered compaction inside a wavefront's words (UN words of loads in flight at a time); wavefront w's supporters
        // follow those of the wavefronts before it.  Any fixed order will do: it only sets the order of the sums.
        int base_out = my_offset;
        for (int j0 = wave; j0 < n_words; j0 += RW * UN) {
            double px[UN], py[UN], pz[UN];
            UV ou[UN], ov[UN];
            bool ok[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int j = j0 + u * RW, i = j * 64 + lane;
                ok[u] = j < n_words && i < n && ((mw[j < n_words ? j : 0] >> lane) & 1ull);
                if (ok[u]) {
                    px[u] = Xs[i * 3], py[u] = Xs[i * 3 + 1], pz[u] = Xs[i * 3 + 2];
                    const XY4 o = obs[i];
                    ou[u] = o.x, ov[u] = o.y;
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const unsigned long long bal = __ballot(ok[u]);
                if (ok[u]) {
                    const int o = base_out + __popcll(bal & ((1ull << lane) - 1ull));
                    sx[o] = px[u], sy[o] = py[u], sz[o] = pz[u];
                    su[o] = ou[u], sv[o] = ov[u];
                }
                base_out += __popcll(bal);
            }
        }
        __syncthreads();
    }

    // body(x, y, z, u, v) for every supporter; every thread adds its points in a fixed order, so a call's sums are
    // reproducible bit for bit
    auto for_supporters = [&](auto body) {
        if (in_lds) {
            for (int i = tid; i < supp; i += RT) body(sx[i], sy[i], sz[i], (double)su[i], (double)sv[i]);
        } else {
            // walk the tracked points UN workgroup-steps at a time with the loads of all of them issued before the
            // arithmetic: this path is latency bound
            for (int base = 0; base < n; base += RT * UN) {
                double px[UN], py[UN], pz[UN], ou[UN], ov[UN];
                bool ok[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int i = base + u * RT + tid;
                    ok[u] = i < n && ((mw[i >> 6] >> (i & 63)) & 1ull);
                    if (ok[u]) {
                        px[u] = Xs[i * 3], py[u] = Xs[i * 3 + 1], pz[u] = Xs[i * 3 + 2];
                        const XY4 o = obs[i];
                        ou[u] = (double)o.x, ov[u] = (double)o.y;
                    }
                }
#pragma unroll
                for (int u = 0; u < UN; ++u)
                    if (ok[u]) body(px[u], py[u], pz[u], ou[u], ov[u]);
            }
        }
    };

    // v[0..N) summed over the workgroup: butterfly inside each wavefront, then the RW partial sums through LDS, added
    // in wavefront order by every thread (same bits everywhere)
    auto block_sum = [&](auto &v, auto n_vals) {
        constexpr int N = decltype(n_vals)::value;
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
        if (RW == 1) return;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < N; ++k) red[wave][k] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double t = red[0][k];
#pragma unroll
            for (int w = 1; w < RW; ++w) t += red[w][k];
            v[k] = t;
        }
        __syncthreads();               // red is free for the next round
    };

    auto cost_of = [&](const Pose &Q) {
        double c[1] = {0};
        for_supporters([&](double x, double y, double z, double ou, double ov) {
            const double xc = Q.R[0][0] * x + Q.R[0][1] * y + Q.R[0][2] * z + Q.t[0];
            const double yc = Q.R[1][0] * x + Q.R[1][1] * y + Q.R[1][2] * z + Q.t[1];
            const double zc = Q.R[2][0] * x + Q.R[2][1] * y + Q.R[2][2] * z + Q.t[2];
            const double iz = rcp_nr(zc);      // refined hardware reciprocal: the minimiser does not depend on the last ulp
            const double ru = fx * xc * iz + cx - ou, rv = fy * yc * iz + cy - ov;
            c[0] += ru * ru + rv * rv;
        });
        block_sum(c, std::integral_constant<int, 1>{});
        return c[0];
    };

    double lambda = 1e-3;
    double cost = cost_of(P);
    int evals = 0;
    bool done = false;
    while (!done && evals < MAX_EVAL) {
        // normal equations at P: 21 entries of J^T J, then the 6 of J^T r
        double Hg[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) Hg[k] = 0;
        for_supporters([&](double x, double y, double z, double ou, double ov) {
            const double xc = P.R[0][0] * x + P.R[0][1] * y + P.R[0][2] * z + P.t[0];
            const double yc = P.R[1][0] * x + P.R[1][1] * y + P.R[1][2] * z + P.t[1];
            const double zc = P.R[2][0] * x + P.R[2][1] * y + P.R[2][2] * z + P.t[2];
            const double iz = rcp_nr(zc);
            const double ru = fx * xc * iz + cx - ou, rv = fy * yc * iz + cy - ov;
            // d(u,v)/dXc
            const double ux = fx * iz, uz = -fx * xc * iz * iz, vy = fy * iz, vz = -fy * yc * iz * iz;
            // dXc/d(omega) = -[Xc]x, dXc/d(upsilon) = I
            const double Ju[6] = {uz * yc, ux * zc - uz * xc, -ux * yc, ux, 0.0, uz};
            const double Jv[6] = {-vy * zc + vz * yc, -vz * xc, vy * xc, 0.0, vy, vz};
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int b = a; b < 6; ++b) Hg[k++] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
                Hg[21 + a] += Ju[a] * ru + Jv[a] * rv;
            }
        });
        block_sum(Hg, std::integral_constant<int, 27>{});

        bool stepped = false;
        while (!stepped && evals < MAX_EVAL) {
            double A[6][6], rhs[6], d[6];
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int b = a; b < 6; ++b) {
                    A[a][b] = Hg[k];
                    A[b][a] = Hg[k];
                    ++k;
                }
                rhs[a] = -Hg[21 + a];
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) A[a][a] *= 1.0 + lambda;   // Marquardt scaling, as CvLevMarq
            ++evals;
            if (!chol_solve6(A, rhs, d)) {
                lambda *= 10.0;
                if (lambda > 1e16) done = true, stepped = true;
                continue;
            }
            const Pose Q = apply_step(P, d);
            const double c2 = cost_of(Q);
            double dn = 0, pn = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) dn += d[a] * d[a];
            pn = 1.0 + P.t[0] * P.t[0] + P.t[1] * P.t[1] + P.t[2] * P.t[2];
            const bool light = lambda <= 1.0;          // the step just tried is within a factor 2 of the Gauss-Newton step
            if (c2 <= cost) {
                P = Q;
                const bool tiny = dn <= (light ? SLAM_REFIT_STEP2 : 1e-18) * pn || (cost - c2) <= 1e-15 * cost;
                cost = c2;
                lambda = fmax(lambda * 0.1, 1e-16);
                stepped = true;
                if (tiny) done = true;
            } else {
                lambda *= 10.0;
                if (lambda > 1e16 || dn <= (light ? SLAM_REFIT_REJECT_STEP2 : 1e-30) * pn) done = true, stepped = true;
            }
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) pose_out[(size_t)seg * 12 + i * 4 + j] = P.R[i][j];
            pose_out[(size_t)seg * 12 + i * 4 + 3] = P.t[i];
        }
#ifdef SLAM_REFIT_PROBE
        status[seg] = (uint8_t)(evals > 0 ? evals : 1);
#else
        status[seg] = 1;
#endif
    }
}

// A pair's pose must not depend on the launch it is part of: slam_track_pairs refits once per chunk, rank block or
// 65535-pair block, and the shard / chunk / multi-GPU paths promise results bit-identical to one whole-shard call.  The
// sums' order is fixed by (wavefronts per pair, LDS path or global walk), so both are functions of the pair's own
// layout only -- never of n_seg.  (Round 2 chose 4 wavefronts below 768 pairs and 1 above, and a larger LDS capacity
// below 256 pairs: the same pair got different pose bits at --chunks >= 5 or on 8 ranks than in the one-GPU run.)
//
// Wavefronts per pair: 4.  On a whole shard one wavefront per pair is ~15 % faster (0.41 against 0.47 ms of a 15.5 ms
// step); four shorten the dependent chain where the launch does not fill the chip (a rank's share of a strongly scaled
// movie, a loop-closure call, the 8k-feature configuration), which is where the refit is exposed.
#ifndef SLAM_REFIT_WAVES
#define SLAM_REFIT_WAVES 4
#endif
// supporters per pair held in LDS (32 bytes each with float32 observations): rows of up to 2048 key-points keep 1536
// (48 KB, three pairs per CU); longer rows (the 8k-feature configuration) up to 4096 (128 KB, one pair per CU)
#ifndef SLAM_REFIT_CAP
#define SLAM_REFIT_CAP 1536
#endif

template <typename XY4, int RW>
int launch_refit_rw(const Mat33 &K, const double *d_X, const XY4 *d_xy1, const int32_t *d_count, int n_seg, int seg_stride,
                    const uint64_t *d_mask_words, const double *d_pose0, double *d_pose, uint8_t *d_status, hipStream_t st,
                    const int32_t *d_gate) {
    const size_t per = 24 + 2 * sizeof(decltype(XY4::x));       // LDS bytes per supporter
    constexpr size_t LDS_DEFAULT = 60 * 1024, LDS_RAISED = 158 * 1024;   // dynamic part; the kernel's static arrays take < 1 KB of the 64 / 160 KB
    int cap = std::min(seg_stride, seg_stride <= 2048 ? SLAM_REFIT_CAP : 4096);
    cap = std::min(cap, (int)(LDS_RAISED / per));
    size_t lds_bytes = (size_t)cap * per;
    if (lds_bytes > LDS_DEFAULT) {
        // once per process and kernel instantiation (one process per GPU); a function-local static is initialised
        // under the runtime's guard, so concurrent first calls from several host threads are safe
        static const bool raise_ok = [] {
            const bool ok = hipFuncSetAttribute((const void *)refit_kernel<XY4, RW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)LDS_RAISED) == hipSuccess;
            (void)hipGetLastError();
            return ok;
        }();
        if (!raise_ok) {
            cap = (int)(LDS_DEFAULT / per);
            lds_bytes = (size_t)cap * per;
        }
    }
    hipLaunchKernelGGL((refit_kernel<XY4, RW>), dim3(n_seg), dim3(RW * 64), lds_bytes, st, K, d_X, d_xy1, d_count, seg_stride,
                       (const unsigned long long *)d_mask_words, d_pose0, d_pose, d_status, cap, d_gate);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

template <typename XY4>
int launch_refit_t(const Mat33 &K, const double *d_X, const XY4 *d_xy1, const int32_t *d_count, int n_seg, int seg_stride,
                   const uint64_t *d_mask_words, const double *d_pose0, double *d_pose, uint8_t *d_status, hipStream_t st,
                    const int32_t *d_gate) {
    return launch_refit_rw<XY4, SLAM_REFIT_WAVES>(K, d_X, d_xy1, d_count, n_seg, seg_stride, d_mask_words, d_pose0, d_pose, d_status, st, d_gate);
}

}  // namespace

int slam_launch_refit(const Mat33 &K, const double *d_X, const void *d_xy1, int xy_f64, const int32_t *d_count, int n_seg,
                      int seg_stride, const uint64_t *d_mask_words, const double *d_pose0, double *d_pose,
                      uint8_t *d_status, hipStream_t st, const int32_t *d_gate) {
    if (n_seg <= 0) return SLAM_OK;
    SLAM_REQUIRE(seg_stride % 64 == 0, "slam_refit_pnp: seg_stride must be a multiple of 64");
    if (xy_f64)
        return launch_refit_t(K, d_X, (const double4 *)d_xy1, d_count, n_seg, seg_stride, d_mask_words, d_pose0, d_pose, d_status, st, d_gate);
    return launch_refit_t(K, d_X, (const float4 *)d_xy1, d_count, n_seg, seg_stride, d_mask_words, d_pose0, d_pose, d_status, st, d_gate);
}

extern "C" int slam_refit_pnp(const double *h_K, const double *d_X, const void *d_xy1, int xy_f64, const int32_t *d_count,
                              int n_seg, int seg_stride, const uint64_t *d_mask_words, const double *d_pose0,
                              double *d_pose, uint8_t *d_status, void *stream) {
    SLAM_REQUIRE(h_K && d_X && d_xy1 && d_count && d_mask_words && d_pose0 && d_pose && d_status,
                 "slam_refit_pnp: null pointer");
    Mat33 K;
    for (int i = 0; i < 9; ++i) K.m[i] = h_K[i];
    return slam_launch_refit(K, d_X, d_xy1, xy_f64, d_count, n_seg, seg_stride, d_mask_words, d_pose0, d_pose, d_status,
                             (hipStream_t)stream);
}

// Rectified-stereo rejection + 4-view join + coordinate gather, one workgroup per frame pair.
//
// Replaces rectified_stereo_pattern_rej (utils/utills.py:168-183), create_rec_dic (:423-435),
// find_kpts_in_all_4_rec (:438-471), tracking_4images (:945-960) and the coordinate gathers of
// get_feature_obj / get_features_{left,right}_coor (:963-997).
//
// The reference builds two dicts {left key-point -> index in the stereo match list} for matches with
// |y_right - y_left| <= REC_ERR and walks the left0->left1 matches in query order.  Every left key-point
// appears at most once in a cross-check match list, so the dicts are the stereo key arrays themselves
// and the walk is an order-preserving compaction over left0 key-point indices: a ballot/prefix scan.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int JB = 256;
constexpr unsigned IDX_MASK = SLAM_MAX_ROWS - 1;

// JU left key-points per thread and pass: their four dependent gathers (stereo key -> right point -> next frame's stereo
// key -> its right point) are in flight together, one memory round trip per level for 2048 key-points instead of one per
// 256, and one barrier pair per pass.
constexpr int JU = 8;

__global__ __launch_bounds__(JB) void rect_join_kernel(const float2 *xy, const int *count, int n_max,
                                                       const unsigned *stereo, const unsigned *temporal,
                                                       const int *pf0, const int *pf1, double rec_err, int *tcount,
                                                       int4 *kp, float4 *xy0, float4 *xy1, int *aux) {
    __shared__ int cnt[JU][JB / 64];          // tracked correspondences per (sub-pass, wave)
    __shared__ int cnt_aux[2][JB / 64];
    const int p = blockIdx.x;
    const int f0 = pf0[p], f1 = pf1[p];
    const int nl0 = min(max(count[2 * f0], 0), n_max);      // caller's count, clamped to the image slot
    const float2 *xl0 = xy + (size_t)(2 * f0) * n_max, *xr0 = xy + (size_t)(2 * f0 + 1) * n_max;
    const float2 *xl1 = xy + (size_t)(2 * f1) * n_max, *xr1 = xy + (size_t)(2 * f1 + 1) * n_max;
    const unsigned *s0 = stereo + (size_t)f0 * n_max, *s1 = stereo + (size_t)f1 * n_max;
    const unsigned *tm = temporal + (size_t)p * n_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0, n_stereo = 0, n_temporal = 0;

    for (int q0 = 0; q0 < nl0; q0 += JB * JU) {
        // key-point of sub-pass u: q0 + u * JB + thread, so ascending (u, wave, lane) is ascending key-point order
        bool valid[JU];
        int r0[JU], t[JU], r1[JU];
        float2 a[JU], b[JU], c[JU], d[JU];
        unsigned k0[JU], kt[JU], k1[JU];
        int ns = 0, nt = 0;
#pragma unroll
        for (int u = 0; u < JU; ++u) {
            const int q = q0 + u * JB + threadIdx.x;
            k0[u] = kt[u] = SLAM_KEY_NONE;
            if (q < nl0) {
                k0[u] = s0[q];
                kt[u] = tm[q];
            }
        }
#pragma unroll
        for (int u = 0; u < JU; ++u) {
            const int q = q0 + u * JB + threadIdx.x;
            const bool has_s = k0[u] != SLAM_KEY_NONE, has_t = kt[u] != SLAM_KEY_NONE;
            ns += __popcll(__ballot(has_s));
            nt += __popcll(__ballot(has_t));
            valid[u] = has_s && has_t;
            r0[u] = k0[u] & IDX_MASK;
            t[u] = kt[u] & IDX_MASK;
            k1[u] = SLAM_KEY_NONE;
            a[u] = b[u] = make_float2(0, 0);
            if (valid[u]) {
                a[u] = xl0[q];
                b[u] = xr0[r0[u]];
                k1[u] = s1[t[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < JU; ++u) {
            // float32 key-point rows widened to float64, |dy| <= REC_ERR inclusive (utills.py:178)
            valid[u] = valid[u] && fabs((double)b[u].y - (double)a[u].y) <= rec_err && k1[u] != SLAM_KEY_NONE;
            r1[u] = k1[u] & IDX_MASK;
            c[u] = d[u] = make_float2(0, 0);
            if (valid[u]) {
                c[u] = xl1[t[u]];
                d[u] = xr1[r1[u]];
            }
        }
        unsigned long long bal[JU];
#pragma unroll
        for (int u = 0; u < JU; ++u) {
            valid[u] = valid[u] && fabs((double)d[u].y - (double)c[u].y) <= rec_err;
            bal[u] = __ballot(valid[u]);
            if (lane == 0) cnt[u][wave] = __popcll(bal[u]);
        }
        if (lane == 0) {
            cnt_aux[0][wave] = ns;
            cnt_aux[1][wave] = nt;
        }
        __syncthreads();
        int off = base;
#pragma unroll
        for (int u = 0; u < JU; ++u) {
#pragma unroll
            for (int w = 0; w < JB / 64; ++w) {
                const int n = cnt[u][w];
                if (valid[u]) {
                    if (w == wave) {
                        const int o = off + __popcll(bal[u] & ((1ull << lane) - 1ull));
                        const size_t g = (size_t)p * n_max + o;
                        kp[g] = make_int4(q0 + u * JB + threadIdx.x, r0[u], t[u], r1[u]);
                        xy0[g] = make_float4(a[u].x, a[u].y, b[u].x, b[u].y);
                        xy1[g] = make_float4(c[u].x, c[u].y, d[u].x, d[u].y);
                    }
                }
                off += n;
            }
        }
#pragma unroll
        for (int w = 0; w < JB / 64; ++w) {
            n_stereo += cnt_aux[0][w];
            n_temporal += cnt_aux[1][w];
        }
        base = off;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tcount[p] = base;
        if (aux) {
            aux[2 * p] = n_stereo;
            aux[2 * p + 1] = n_temporal;
        }
    }
}

}  // namespace

int slam_launch_rect_join(const float *d_xy, const int32_t *d_count, int n_max, const uint32_t *d_stereo_key,
                          const uint32_t *d_temporal_key, const int32_t *d_pair_f0, const int32_t *d_pair_f1,
                          int n_pairs, double rec_err, int32_t *d_tracked_count, int32_t *d_tracked_kp, float *d_xy0,
                          float *d_xy1, int32_t *d_aux, hipStream_t st) {
    if (n_pairs <= 0) return SLAM_OK;
    SLAM_REQUIRE(d_xy && d_count && d_stereo_key && d_temporal_key && d_pair_f0 && d_pair_f1 && d_tracked_count &&
                     d_tracked_kp && d_xy0 && d_xy1,
                 "slam_rect_filter_join: null pointer");
    SLAM_REQUIRE(n_max > 0 && n_max < SLAM_MAX_ROWS, "slam_rect_filter_join: n_max %d out of range", n_max);
    hipLaunchKernelGGL(rect_join_kernel, dim3(n_pairs), dim3(JB), 0, st, (const float2 *)d_xy, d_count, n_max,
                       d_stereo_key, d_temporal_key, d_pair_f0, d_pair_f1, rec_err, d_tracked_count,
                       (int4 *)d_tracked_kp, (float4 *)d_xy0, (float4 *)d_xy1, d_aux);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_rect_filter_join(const float *d_xy, const int32_t *d_count, int n_max,
                                     const uint32_t *d_stereo_key, const uint32_t *d_temporal_key,
                                     const int32_t *d_pair_f0, const int32_t *d_pair_f1, int n_pairs, double rec_err,
                                     int32_t *d_tracked_count, int32_t *d_tracked_kp, float *d_xy0, float *d_xy1,
                                     void *stream) {
    return slam_launch_rect_join(d_xy, d_count, n_max, d_stereo_key, d_temporal_key, d_pair_f0, d_pair_f1, n_pairs,
                                 rec_err, d_tracked_count, d_tracked_kp, d_xy0, d_xy1, nullptr, (hipStream_t)stream);
}

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

__global__ __launch_bounds__(JB) void rect_join_kernel(const float2 *xy, const int *count, int n_max,
                                                       const unsigned *stereo, const unsigned *temporal,
                                                       const int *pf0, const int *pf1, double rec_err, int *tcount,
                                                       int4 *kp, float4 *xy0, float4 *xy1, int *aux) {
    __shared__ int wave_tot[JB / 64];
    __shared__ int wave_aux[2][JB / 64];
    const int p = blockIdx.x;
    const int f0 = pf0[p], f1 = pf1[p];
    const int nl0 = count[2 * f0];
    const float2 *xl0 = xy + (size_t)(2 * f0) * n_max, *xr0 = xy + (size_t)(2 * f0 + 1) * n_max;
    const float2 *xl1 = xy + (size_t)(2 * f1) * n_max, *xr1 = xy + (size_t)(2 * f1 + 1) * n_max;
    const unsigned *s0 = stereo + (size_t)f0 * n_max, *s1 = stereo + (size_t)f1 * n_max;
    const unsigned *tm = temporal + (size_t)p * n_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0, n_stereo = 0, n_temporal = 0;

    for (int q0 = 0; q0 < nl0; q0 += JB) {
        const int q = q0 + threadIdx.x;
        bool valid = false, has_s = false, has_t = false;
        int r0 = 0, t = 0, r1 = 0;
        float2 a = {0, 0}, b = {0, 0}, c = {0, 0}, d = {0, 0};
        if (q < nl0) {
            const unsigned k0 = s0[q], kt = tm[q];
            has_s = k0 != SLAM_KEY_NONE;
            has_t = kt != SLAM_KEY_NONE;
            if (has_s && has_t) {
                r0 = k0 & IDX_MASK;
                t = kt & IDX_MASK;
                a = xl0[q];
                b = xr0[r0];
                // float32 key-point rows widened to float64, |dy| <= REC_ERR inclusive (utills.py:178)
                if (fabs((double)b.y - (double)a.y) <= rec_err) {
                    const unsigned k1 = s1[t];
                    if (k1 != SLAM_KEY_NONE) {
                        r1 = k1 & IDX_MASK;
                        c = xl1[t];
                        d = xr1[r1];
                        valid = fabs((double)d.y - (double)c.y) <= rec_err;
                    }
                }
            }
        }
        const unsigned long long bal = __ballot(valid);
        const unsigned long long bs = __ballot(has_s), bt = __ballot(has_t);
        if (lane == 0) {
            wave_tot[wave] = __popcll(bal);
            wave_aux[0][wave] = __popcll(bs);
            wave_aux[1][wave] = __popcll(bt);
        }
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int w = 0; w < JB / 64; ++w) {
            if (w < wave) off += wave_tot[w];
            tot += wave_tot[w];
            n_stereo += wave_aux[0][w];
            n_temporal += wave_aux[1][w];
        }
        if (valid) {
            const int o = off + __popcll(bal & ((1ull << lane) - 1ull));
            const size_t g = (size_t)p * n_max + o;
            kp[g] = make_int4(q, r0, t, r1);
            xy0[g] = make_float4(a.x, a.y, b.x, b.y);
            xy1[g] = make_float4(c.x, c.y, d.x, d.y);
        }
        base += tot;
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

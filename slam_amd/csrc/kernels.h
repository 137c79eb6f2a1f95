// Internal launchers shared between the per-stage C ABI (one file per stage) and the shard pipeline.
#pragma once
#include "common.h"

// slam_match_hamming_batched with an optional launch order: d_order[k] = the problem the k-th group of workgroups
// takes (a permutation of 0 .. n_prob-1; NULL = identity).  Outputs are indexed by problem and do not depend on it.
int slam_launch_match_batched(const uint8_t *d_desc, const int32_t *d_count, int n_max, const int32_t *d_prob_query,
                              const int32_t *d_prob_train, const int32_t *d_order, int n_prob, int norm, int mode,
                              double ratio, uint32_t *d_key1, uint32_t *d_key2, void *d_workspace, size_t workspace_bytes,
                              hipStream_t st);

int slam_launch_rect_join(const float *d_xy, const int32_t *d_count, int n_max, const uint32_t *d_stereo_key,
                          const uint32_t *d_temporal_key, const int32_t *d_pair_f0, const int32_t *d_pair_f1,
                          int n_pairs, double rec_err, int32_t *d_tracked_count, int32_t *d_tracked_kp, float *d_xy0,
                          float *d_xy1, int32_t *d_aux, hipStream_t st);

// np.random.seed(d_seed[seg]); draws [d_begin[row], d_begin[row] + n_draw) of np.random.randint(0, d_count[seg], 4),
// generated on the device (MT19937 + masked rejection), into d_sample[row][n_draw*4].  d_begin == NULL: from draw 0.
// d_produced (optional): [row] = n_draw where the segment has points, else 0.
int slam_launch_draw_samples(const uint32_t *d_seed, const int32_t *d_count, const int32_t *d_seg_of,
                             const int32_t *d_begin, int n_rows, int n_draw, int32_t *d_sample, int32_t *d_produced,
                             hipStream_t st);

// Hypotheses + scoring with an optional row -> segment indirection (rows = active pairs of a RANSAC round).
int slam_launch_p3p(const Mat33 &K, const double *d_X, const void *d_xy1, int xy_f64, const int32_t *d_count,
                    const int32_t *d_seg_of, int n_rows, int seg_stride, const int32_t *d_sample, int sample_stride,
                    int draw0, int n_hyp, double *d_pose, uint8_t *d_status, hipStream_t st);
int slam_launch_score(const Mat33 &K, const Mat34 &M2, const double *d_X, const void *d_xy1, int xy_f64, const int32_t *d_count,
                      const int32_t *d_seg_of, int n_rows, int seg_stride, const double *d_pose,
                      const uint8_t *d_status, int n_hyp, double thr, int32_t *d_inlier_count, uint64_t *d_mask,
                      hipStream_t st, const int32_t *d_gate = nullptr, const float *d_packed = nullptr,
                      const float *d_scale = nullptr);
// d_packed / d_scale (optional, counting on a rectified rig): the points' single-precision records written by
// slam_launch_pack_points for the same d_X / d_xy1 / d_count -- [n_seg][seg_stride][8] floats and [n_seg] floats; the
// counts are the same, the scoring kernels classify in single precision first (score.hip).
int slam_launch_pack_points(const double *d_X, const float *d_xy1, const int32_t *d_count, int n_seg, int seg_stride,
                            float *d_packed, float *d_scale, hipStream_t st);
int slam_launch_refit(const Mat33 &K, const double *d_X, const void *d_xy1, int xy_f64, const int32_t *d_count, int n_seg,
                      int seg_stride, const uint64_t *d_mask_words, const double *d_pose0, double *d_pose,
                      uint8_t *d_status, hipStream_t st, const int32_t *d_gate = nullptr);
// d_gate (both launchers above; optional): device word read when the kernel starts -- unless it is zero the kernel
// returns without touching its outputs (work queued speculatively behind a device-side decision, pipeline.hip).
int slam_launch_triangulate(const Mat34 &P, const Mat34 &Q, const float *d_xy, const int32_t *d_count, int n_seg,
                            int seg_stride, double *d_X, hipStream_t st);

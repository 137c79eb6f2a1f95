/* slam_amd.h -- C ABI of libslam_amd.so: the MI355X (gfx950) implementation of the stereo-trajectory
 * hot path of maoratar1/SLAM (utils/utills.py).
 *
 * The reference is pure Python and has no FFI layer; its boundary for this path is the set of
 * module-level functions of utils/utills.py.  Each entry point below names the reference
 * function(s) whose arithmetic it replaces (paths relative to the reference root).
 * slam_amd/utills.py binds them (ctypes) behind the reference's own signatures; INTEGRATION.md
 * shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative SLAM_E_* code; slam_last_error_string()
 *     describes the last failure on the calling thread.  No exception crosses the ABI.
 *   - `d_` pointers are device memory owned by the caller; `h_` pointers are host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous on that stream unless stated otherwise.  No entry point allocates or frees
 *     device memory: scratch comes from the caller-provided workspace.
 *   - descriptors are rows of SLAM_DESC_STRIDE (64) bytes: the 61 bytes of an AKAZE MLDB descriptor
 *     followed by zero padding (zero padding is distance-neutral).
 *   - a packed match key is (distance << 20) | index, 0xFFFFFFFF = no match.
 */
#ifndef SLAM_AMD_H
#define SLAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_DESC_STRIDE 64
#define SLAM_KEY_SHIFT 20
#define SLAM_KEY_NONE 0xFFFFFFFFu
#define SLAM_MAX_ROWS (1 << SLAM_KEY_SHIFT)

/* cv2 enum values the reference passes (utils/utills.py:24) */
#define SLAM_NORM_HAMMING 6
#define SLAM_NORM_HAMMING2 7

/* matcher modes: utils/utills.py:108-110 (cross-check brute force) and :112-119 (kNN, k = 2).
 * cv2.BFMatcher(crossCheck=True) has had two behaviours and the reference pins no OpenCV version:
 *   SLAM_MATCH_CROSSCHECK         batchDistance(crosscheck=true) as in OpenCV up to 4.5: every TRAIN row votes for its
 *                                 nearest query; a query keeps the voter of least distance (lowest train row on ties).
 *                                 It does not check that the train row is the query's own nearest.  The default.
 *   SLAM_MATCH_CROSSCHECK_MUTUAL  newer OpenCV: both directions are computed and a pair is kept only if each row is
 *                                 the other's nearest (first minimum on both sides).  Twice the distance work. */
#define SLAM_MATCH_CROSSCHECK 0
#define SLAM_MATCH_KNN2 1
#define SLAM_MATCH_CROSSCHECK_MUTUAL 2

#define SLAM_OK 0
#define SLAM_E_INVALID (-1)     /* bad argument (null pointer, size out of range) */
#define SLAM_E_WORKSPACE (-2)   /* workspace too small */
#define SLAM_E_HIP (-3)         /* a HIP runtime call failed */
#define SLAM_E_EMPTY (-4)       /* empty tracked set: the reference raises ValueError (utills.py:502) */
#define SLAM_E_DUPLICATE (-5)   /* slam_chain_tracks: a key-point appears twice inside one pair */
#define SLAM_E_HOST (-6)        /* host-side failure inside the library (out of host memory, a worker thread failed) */

const char *slam_version(void);
const char *slam_last_error_string(void);
/* Explicit teardown (no counterpart in the reference, which owns no threads and no device objects): stops and joins the
 * library's host worker pool and releases the calling thread's internal streams / events while the HIP runtime is still
 * up.  Optional -- nothing in the library joins threads or calls HIP from a static destructor -- and idempotent; the
 * library stays usable afterwards (host-side steps then run on the calling thread only).  slam_amd/_lib.py calls it
 * from a Python atexit hook, i.e. before any library's exit handlers run. */
int slam_shutdown(void);

/* ---------------------------------------------------------------------------------------------
 * Matching.  Replaces cv2.BFMatcher(NORM_HAMMING2, crossCheck=True).match (utils/utills.py:57-70,
 * :105-110) and cv2.BFMatcher(...).knnMatch(.., 2) (utils/utills.py:92-102).
 *
 * SLAM_MATCH_CROSSCHECK / _MUTUAL: d_key1[i] = packed (distance, train row) matched to query row i under
 *   that cross-check rule, or SLAM_KEY_NONE; d_key2 is not written (may be NULL).
 * SLAM_MATCH_KNN2: d_key1[i] / d_key2[i] = nearest / second nearest train row of query i (ties:
 *   lower train index first).
 * d_query / d_train: [n][64] bytes, 16-byte aligned.  n_query, n_train < SLAM_MAX_ROWS.
 * Workspace: slam_match_workspace_bytes(n_query, n_train) bytes of device memory.
 */
size_t slam_match_workspace_bytes(int n_query, int n_train);
int slam_match_hamming(const uint8_t *d_query, int n_query, const uint8_t *d_train, int n_train, int norm,
                       int mode, uint32_t *d_key1, uint32_t *d_key2, void *d_workspace, size_t workspace_bytes,
                       void *stream);

/* slam_match_hamming for callers that cannot allocate device memory and have no workspace to bring: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference (which
 * never touches a GPU); it exists only for those callers.  Same inputs, outputs and results as slam_match_hamming; the
 * per-split partial lists live in a static device array inside the library (4.5 MiB per GPU, held until the process
 * ends) instead of a caller workspace, so n_query and n_train must each be <= SLAM_MATCH_NOALLOC_MAX_ROWS (larger
 * sizes return SLAM_E_INVALID and launch nothing).  The call allocates and frees nothing (no hipMalloc / hipFree, no
 * allocating library routine).  The device that owns `stream` must be current.  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (different streams or threads) race; serialise them. */
#define SLAM_MATCH_NOALLOC_MAX_ROWS 8192
int slam_match_hamming_noalloc(const uint8_t *d_query, int n_query, const uint8_t *d_train, int n_train, int norm,
                               int mode, uint32_t *d_key1, uint32_t *d_key2, void *stream);

/* Batched form: one launch for many image pairs.  Image k holds h/d_count[k] rows at
 * d_desc + k * n_max * 64.  Problem p matches query image prob_query[p] against train image
 * prob_train[p]; results go to d_key1 + p * n_max (and d_key2 + p * n_max for KNN2): all n_max keys of every problem
 * are written (SLAM_KEY_NONE beyond the query image's rows).
 * d_prob_query / d_prob_train / d_count are device int32 arrays.
 * ratio (KNN2 only): > 0 applies significance_test (utils/utills.py:131-152) on the device: d_key1[i] is kept only if
 *   distance1 < ratio * distance2 in float64 (RATIO = 0.6, utils/utills.py:23); 0 = no test.
 * Workspace: slam_match_batched_workspace_bytes(n_prob, n_max, mode) bytes of device memory (per-row nearest keys
 * before the rule is applied; a batch too small to fill the chip also splits the scanned image over workgroups). */
size_t slam_match_batched_workspace_bytes(int n_prob, int n_max, int mode);
int slam_match_hamming_batched(const uint8_t *d_desc, const int32_t *d_count, int n_max, const int32_t *d_prob_query,
                               const int32_t *d_prob_train, int n_prob, int norm, int mode, double ratio,
                               uint32_t *d_key1, uint32_t *d_key2, void *d_workspace, size_t workspace_bytes,
                               void *stream);

/* Unpacks keys into the (trainIdx, distance) pairs of cv2.DMatch; -1 / INT32_MAX where there is no match. */
int slam_unpack_keys(const uint32_t *d_key, int n, int32_t *d_index, int32_t *d_distance, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Rectified-stereo rejection + 4-view join + coordinate gather.  Replaces
 * rectified_stereo_pattern_rej (utils/utills.py:168-183), create_rec_dic (:423-435),
 * find_kpts_in_all_4_rec (:438-471), tracking_4images (:945-960) and the gathers of
 * get_feature_obj / get_features_*_coor (:963-997), batched over frame pairs.
 *
 * Frame f owns images 2f (left) and 2f+1 (right); d_xy is [n_img][n_max][2] float32.
 * d_stereo_key: [n_frames][n_max] cross-check keys, query = left image, train = right image.
 * d_temporal_key: [n_pairs][n_max] cross-check keys, query = left of pair_f0[p], train = left of pair_f1[p].
 * Outputs per pair p (rows beyond d_tracked_count[p] are untouched):
 *   d_tracked_count[p]             number T of correspondences seen in all four images
 *   d_tracked_kp[p][n_max][4]      key-point indices (left0, right0, left1, right1), ascending left0
 *   d_xy0[p][n_max][4]             float32 (xL0, yL0, xR0, yR0)
 *   d_xy1[p][n_max][4]             float32 (xL1, yL1, xR1, yR1)
 * rec_err: REC_ERR (utils/utills.py:41), inclusive, compared in float64.
 */
int slam_rect_filter_join(const float *d_xy, const int32_t *d_count, int n_max, const uint32_t *d_stereo_key,
                          const uint32_t *d_temporal_key, const int32_t *d_pair_f0, const int32_t *d_pair_f1,
                          int n_pairs, double rec_err, int32_t *d_tracked_count, int32_t *d_tracked_kp,
                          float *d_xy0, float *d_xy1, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Triangulation.  Replaces linear_least_square + triangulate (utils/utills.py:220-255): per
 * correspondence the 4x4 DLT matrix, its right singular vector of least singular value,
 * dehomogenised.  h_P / h_Q: row-major 3x4 float64 projection matrices (K@M1, K@M2).
 * Batched: segment s holds d_count[s] points at d_xy + s*seg_stride*4 (float32 xL,yL,xR,yR);
 * output d_X + s*seg_stride*3 (float64).  d_count == NULL means one segment of seg_stride points.
 */
int slam_triangulate(const double *h_P, const double *h_Q, const float *d_xy, const int32_t *d_count, int n_seg,
                     int seg_stride, double *d_X, void *stream);
/* float64 coordinates, single segment (the drop-in `triangulate` receives float64 arrays) */
int slam_triangulate_f64(const double *h_P, const double *h_Q, const double *d_xy_l, const double *d_xy_r, int n,
                         double *d_X, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Minimal-sample index streams.  Reproduces np.random.seed(seed); np.random.randint(0, n, 4)
 * repeated (utils/utills.py:502): MT19937 + numpy's masked rejection.
 * slam_mt19937_words: host; fills h_words[n_words] with the raw 32-bit outputs after seeding.
 * slam_randint_host: host; h_out[n_out] = the first n_out values of randint(0, bound, ...);
 *   returns the number of raw words consumed (>= 0) or a negative error.
 */
int slam_mt19937_words(uint32_t seed, uint32_t *h_words, int n_words);
long slam_randint_host(uint32_t seed, uint32_t bound, int32_t *h_out, long n_out);

/* The bookkeeping of the adaptive loop of online_est_pnp_ransac (utils/utills.py:488-491, :498-523), replayed on the
 * host over hypotheses that were solved and scored in batches: for each draw in [draw_begin, draw_end), in order -- stop
 * test (outliers_perc != 0, it < log(1 - prob) / log(1 - (1 - outliers_perc)^4), it < max_iterations; or
 * it < fixed_iterations when that is > 0), a failed solve consumes the draw without counting (:506-507), strict-> best
 * (:514), cumulative inlier / outlier totals and the clipped outlier ratio (:521-523).  h_status / h_count are indexed by
 * (draw - draw_begin).  The state carries over from batch to batch; slam_ransac_state_init sets the loop's start values.
 * Returns 0; st->finished says whether the loop ended inside the batch (st->hit_cap: by the max_draws bound, which the
 * reference does not have), st->draws how many draws it consumed in all.  Host only; the same code slam_track_pairs uses. */
typedef struct SlamRansacState {
    int32_t it, draws, max_supp, best_draw, finished, hit_cap;
    int64_t inliers, outliers;
    double outliers_perc;
    double est_for, est;     /* cache of the iteration estimate */
} SlamRansacState;
int slam_ransac_state_init(SlamRansacState *st);
int slam_ransac_replay(SlamRansacState *st, int n_points, const uint8_t *h_status, const int32_t *h_count, int draw_begin,
                       int draw_end, double prob, double max_iterations, int fixed_iterations, int max_draws);
/* The same stream generated on the device, for many pairs at once: for row r,
 *   np.random.seed(d_seed[r]); draws [d_begin[r], d_begin[r] + n_draw) of repeated np.random.randint(0, d_count[r], 4)
 * -> d_sample[r][n_draw][4] (int32).  Rows with d_count[r] <= 0 are left untouched.  Used by the later RANSAC rounds
 * of slam_track_pairs, whose draw ranges depend on how far each pair's loop (utils/utills.py:498-523) has got. */
int slam_randint_draws(const uint32_t *d_seed, const int32_t *d_count, const int32_t *d_begin, int n_rows, int n_draw,
                       int32_t *d_sample, void *stream);

/* ---------------------------------------------------------------------------------------------
 * PnP hypotheses.  Replaces pnp(..., cv2.SOLVEPNP_AP3P) + rodriguez_to_mat (utils/utills.py:574-596,
 * call site :503): Ke & Roumeliotis AP3P on the first three sample points, the fourth picks the root.
 * Segment s (a frame pair) has d_count[s] points: d_X + s*seg_stride*3 (float64) and
 * d_xy1 + s*seg_stride*4 (xL1,yL1,xR1,yR1; float32 rows, or float64 rows when xy_f64 != 0 -- the drop-in
 * wrappers receive float64 arrays; only the left pair is used).  d_sample holds
 * n_hyp x 4 point indices per segment ([n_seg][n_hyp][4] int32).  Outputs [n_seg][n_hyp]:
 *   d_pose[..][12]  row-major 3x4 [R|t] (NaN-filled where the sample is degenerate)
 *   d_status[..]    1 = pose returned, 0 = solver found no root (the reference's `continue`, :506-507)
 * h_K: row-major 3x3 calibration.
 */
int slam_p3p_hypotheses(const double *h_K, const double *d_X, const void *d_xy1, int xy_f64, const int32_t *d_count,
                        int n_seg, int seg_stride, const int32_t *d_sample, int n_hyp, double *d_pose,
                        uint8_t *d_status, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Consensus scoring.  Replaces consensus_match -> compose_transformations, find_supporters_4_frame,
 * project, check_projection, compute_square_dist (utils/utills.py:533-545, :659-736): for every
 * hypothesis, project all points of its segment into left1 and right1 and test the un-squared
 * pixel distance against thr = acc**2 (inclusive) in both.
 * d_pose: [n_seg][n_hyp][12]; d_status as above (status 0 hypotheses are skipped, count 0).
 * d_inlier_count: [n_seg][n_hyp] int32.  d_mask (optional, may be NULL): [n_seg][n_hyp][seg_stride/64]
 * uint64 bit masks, bit i%64 of word i/64 = point i supports.  seg_stride must be a multiple of 64.
 * h_K 3x3, h_M2 3x4 row-major float64.
 */
int slam_score_hypotheses(const double *h_K, const double *h_M2, const double *d_X, const void *d_xy1, int xy_f64,
                          const int32_t *d_count, int n_seg, int seg_stride, const double *d_pose,
                          const uint8_t *d_status, int n_hyp, double thr, int32_t *d_inlier_count, uint64_t *d_mask,
                          void *stream);
/* The same counts with a single-precision pre-classification (rectified rigs: K = [fx 0 cx; 0 fy cy; 0 0 1],
 * M2 = [I | (tx, 0, 0)]; any other rig, or thr outside [1e-3, 1e6], takes the kernels of slam_score_hypotheses).
 * A caller that scores many hypothesis batches against the same points (the RANSAC rounds of slam_track_pairs do)
 * converts them once with slam_score_pack_points -- d_packed float32 [n_seg][seg_stride][8] (16-byte aligned),
 * d_scale float32 [n_seg] -- and passes both here next to the float64 points and float32 observations they were made
 * from.  A test is decided in single precision only when it is farther from the radius than a rigorous bound on
 * everything single precision can have lost; the rest (about one test in a million) goes through the float64
 * arithmetic of slam_score_hypotheses, so d_inlier_count is identical (score.hip has the error analysis). */
int slam_score_pack_points(const double *d_X, const float *d_xy1, const int32_t *d_count, int n_seg, int seg_stride,
                           float *d_packed, float *d_scale, void *stream);
int slam_score_hypotheses_packed(const double *h_K, const double *h_M2, const double *d_X, const float *d_xy1,
                                 const float *d_packed, const float *d_scale, const int32_t *d_count, int n_seg,
                                 int seg_stride, const double *d_pose, const uint8_t *d_status, int n_hyp, double thr,
                                 int32_t *d_inlier_count, void *stream);
/* float64 observations, one segment, one pose (drop-in `consensus_match`) */
int slam_consensus_match_f64(const double *h_K, const double *h_M2, const double *h_pose, const double *d_X,
                             const double *d_left1, const double *d_right1, int n, double thr, uint8_t *d_mask,
                             void *stream);

/* ---------------------------------------------------------------------------------------------
 * Refit.  Replaces extra_refine -> pnp(..., cv2.SOLVEPNP_ITERATIVE) (utils/utills.py:548-571):
 * Levenberg-Marquardt on the left-image reprojection error over the supporters, started from the
 * winning hypothesis.  Per segment: d_mask_words [seg_stride/64] uint64, d_pose0 [12] -> d_pose [12];
 * d_status: 1 = refit done, 0 = fewer than 6 supporters (pose None, :558-559), NaN-filled pose.
 */
int slam_refit_pnp(const double *h_K, const double *d_X, const void *d_xy1, int xy_f64, const int32_t *d_count, int n_seg,
                   int seg_stride, const uint64_t *d_mask_words, const double *d_pose0, double *d_pose,
                   uint8_t *d_status, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Ingest: dense wire format -> padded device layout.  No counterpart in the reference, which hands every image's
 * detections over as Python objects (utils/utills.py:46-54, cv2 detectAndCompute -> key-point list + descriptor ndarray);
 * this is how a shard of them reaches the layout every entry point above reads (d_desc [n_img][n_max][64],
 * d_xy [n_img][n_max][2]) with 8 % fewer bytes over PCIe than uploading that layout itself.
 *   d_dense_desc [n_rows_total][desc_width]  descriptor rows at their own width (61 for AKAZE MLDB), any image order;
 *                                            4-byte aligned
 *   d_dense_xy   [n_rows_total][2]           KeyPoint.pt of the same rows; 8-byte aligned
 *   d_row_ptr    [n_img] int64               first dense row of image k;  d_count [n_img]: its rows
 * Every byte of d_desc / d_xy is written: rows past an image's count and bytes past desc_width are zero, so the result
 * equals a padded upload.  d_count_out [n_img]: the counts clamped to [0, n_max]; an image whose row range does not lie
 * inside the dense arrays is unpacked as empty.
 */
int slam_unpack_detections(const uint8_t *d_dense_desc, const float *d_dense_xy, const int64_t *d_row_ptr,
                           const int32_t *d_count, int n_img, int n_max, int desc_width, int64_t n_rows_total,
                           uint8_t *d_desc, float *d_xy, int32_t *d_count_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Trajectory composition (host).  Replaces convert_trans_from_rel_to_global (utils/utills.py:749-757):
 * global[i] = rel[i] @ [global[i-1]; 0 0 0 1], started from rel[0] (so rel[0] is applied twice, as in the
 * reference, which is harmless for [I|0]).  h_rel, h_global: [n][12] row-major 3x4.
 */
int slam_compose_global(const double *h_rel, int n, double *h_global);

/* ---------------------------------------------------------------------------------------------
 * Track chaining (host).  Replaces DataBase.create_db / find_tracks_between_consecutive_frame
 * (DataBaseDirectory/DataBase.py:56-153) on the hot path's index arrays: for every supporter of pair (i, i+1), in
 * order, the frame-i LEFT key-point is looked up among the frame-i key-points the previous pair's supporters ended
 * on; a hit extends that track, a miss opens track number n_tracks.  Numbering (encounter order) and the order of the
 * track ids inside every frame are the reference's.
 *   h_kp [M][4]              supporters as key-point indices (L_i, R_i, L_{i+1}, R_{i+1}), pair after pair
 *   h_pair_ptr [n_pairs+1]   supporters of pair i are rows h_pair_ptr[i] .. h_pair_ptr[i+1]; M = h_pair_ptr[n_pairs]
 *   n_kp                     left key-point indices are < n_kp
 *   h_xy                     optional float32 [n_pairs+1][2][n_max][2] key-point coordinates (left, right) per frame
 * Outputs, sized by the caller for the worst case (every supporter opens a track):
 *   h_track_ptr [M+1], h_obs_frame [2M], h_obs_kp [2M][2], h_obs_xy [2M][4] (x_left, x_right, y_left, y_right; only
 *   with h_xy), h_frame_ptr [n_pairs+2], h_frame_tracks [2M]; *h_n_tracks, *h_n_obs say how much of them is used:
 *   observations of track t are rows h_track_ptr[t] .. h_track_ptr[t+1], ascending frame.
 * SLAM_E_DUPLICATE when a left key-point of either frame appears twice inside a pair (the reference's dictionaries
 * would overwrite; the hot path never produces that: the join emits ascending L_i and cross-check is injective).
 */
int slam_chain_tracks(const int32_t *h_kp, const int64_t *h_pair_ptr, int32_t n_pairs, int32_t n_kp, const float *h_xy,
                      int32_t n_max, int64_t *h_track_ptr, int32_t *h_obs_frame, int32_t *h_obs_kp, double *h_obs_xy,
                      int64_t *h_frame_ptr, int64_t *h_frame_tracks, int64_t *h_n_tracks, int64_t *h_n_obs);

/* ---------------------------------------------------------------------------------------------
 * Whole shard.  Replaces the body of find_features_in_consecutive_frames_whole_movie
 * (utils/utills.py:834-873) / find_loop_cand_by_consensus (:1118-1146) minus detection: stereo match
 * of every frame, then per frame pair temporal match, join, triangulation, adaptive PnP-RANSAC
 * (online_est_pnp_ransac, :474-530, replayed on the host over batched device results) and refit.
 * Synchronous: returns when the host outputs are filled.  slam_track_last_timing[7] = number of match launches.
 */
typedef struct SlamTrackConfig {
    double K[9];             /* calibration, row-major */
    double M1[12];           /* left extrinsic [I|0] */
    double M2[12];           /* right-to-left extrinsic */
    double rec_err;          /* REC_ERR = 2 */
    double acc;              /* SUPP_ERR = 2; the test is distance <= acc*acc */
    double prob;             /* 0.9999 */
    double max_iterations;   /* 1e3 */
    int32_t fixed_iterations;/* > 0: ignore the adaptive stop and run exactly this many counted iterations */
    int32_t max_draws;       /* bound on draws incl. failed ones (the reference is unbounded); 0 = 64 * max_iterations */
    int32_t norm;            /* SLAM_NORM_HAMMING2 */
    int32_t first_chunk;     /* hypotheses evaluated speculatively in round 1 (0 = 64) */
    int32_t pipeline_chunks; /* 0: the library chooses (two blocks for a large shard, one for a small one);
                                > 1: split the pair list into this many contiguous blocks and pipeline them over two
                                internal streams (RANSAC rounds of one block overlap the matching of the next);
                                needs pairs that are local in time, otherwise the call falls back to one block */
    int32_t pipeline_taper_permille; /* size of each block relative to the previous one, in 1/1000: 0 or 1000 = equal
                                blocks; e.g. 450 makes every block 0.45 x the one before, so the un-overlapped tail of
                                the call belongs to a small block.  Results do not depend on it. */
    int32_t match_mode;      /* how match() pairs descriptors (utils/utills.py:105-119): SLAM_MATCH_CROSSCHECK (default,
                                MATCH_METHOD = BRUTE_FORCE), SLAM_MATCH_CROSSCHECK_MUTUAL, or SLAM_MATCH_KNN2
                                (REGULAR_KNN: two nearest + the ratio test below) */
    int32_t pipeline_serial_match; /* with pipeline_chunks > 1: nonzero = a block's matcher starts only when the previous
                                block's matcher has finished (the blocks' matchers run back to back, each block's
                                RANSAC / refit beside the NEXT block's matcher); 0 = blocks on different streams start
                                together.  Scheduling only: results do not depend on it. */
    double ratio;            /* RATIO = 0.6 (utils/utills.py:23); used by SLAM_MATCH_KNN2 only; 0 = 0.6 */
} SlamTrackConfig;

typedef struct SlamTrackIO {
    /* device inputs */
    const uint8_t *d_desc;   /* [2*n_frames][n_max][64]: image 2f = left of frame f, 2f+1 = right */
    const float *d_xy;       /* [2*n_frames][n_max][2] */
    const int32_t *d_count;  /* [2*n_frames] */
    int32_t n_frames, n_max;
    /* host inputs */
    const int32_t *h_pair_f0, *h_pair_f1; /* [n_pairs] frame indices (consecutive: f, f+1) */
    const uint32_t *h_pair_seed;          /* [n_pairs] np.random.seed value in force before each pair's RANSAC */
    int32_t n_pairs;
    /* device outputs (caller-allocated) */
    int32_t *d_tracked_kp;   /* [n_pairs][n_max][4] */
    /* host outputs (caller-allocated) */
    /* h_pose and h_mask should be page-locked (hipHostMalloc / hipHostRegister, torch pin_memory): the library then
       queues the result copies ahead of time; with pageable memory it still works, one host round trip slower per block */
    double *h_pose;          /* [n_pairs][12]; NaN where the reference returns None */
    int32_t *h_stats;        /* [n_pairs][8]: tracked T, supporters, iterations, draws, best draw, status, stereo0 matches, temporal matches */
    uint64_t *h_mask;        /* [n_pairs][n_max/64] supporter bit masks (pre-refit, as the reference returns) */
    /* scratch */
    void *d_workspace; size_t d_workspace_bytes;
    void *h_workspace; size_t h_workspace_bytes; /* pinned host memory */
} SlamTrackIO;

/* status bits in h_stats[p][5] */
#define SLAM_PAIR_OK 1          /* refit pose returned */
#define SLAM_PAIR_EMPTY 2       /* T == 0: the reference raises ValueError */
#define SLAM_PAIR_FEW 4         /* < 6 supporters: pose None */
#define SLAM_PAIR_DRAW_CAP 8    /* max_draws reached before the loop condition ended */

size_t slam_track_device_workspace_bytes(int n_frames, int n_pairs, int n_max, const SlamTrackConfig *cfg);
size_t slam_track_host_workspace_bytes(int n_frames, int n_pairs, int n_max, const SlamTrackConfig *cfg);
int slam_track_pairs(const SlamTrackConfig *cfg, const SlamTrackIO *io, void *stream);

/* slam_track_pairs for a caller that already holds the frames' stereo matches.  d_stereo_key: [io->n_frames][io->n_max]
 * device keys, row f = d_key1 of slam_match_hamming_batched for the problem (query image 2f, train image 2f+1) under
 * the same norm / match_mode / ratio.  The call launches only the temporal problems and reads no right-image
 * descriptor; everything else, results included, is slam_track_pairs' (NULL = slam_track_pairs itself).
 * The reference computes a frame's stereo match once per use (utils/utills.py:1126-1134 re-matches every candidate of
 * every current frame); a multi-GPU loop-closure pass matches each frame on ONE rank and all-gathers the keys
 * (slam_amd/dist.py: stereo_keys_sharded), then every rank's pair block comes through here. */
int slam_track_pairs_stereo_keys(const SlamTrackConfig *cfg, const SlamTrackIO *io, const uint32_t *d_stereo_key,
                                 void *stream);

/* per-stage device times (ms) of the last slam_track_pairs call on this thread, measured with HIP events on
 * `stream`: [0] match (time with a match launch in flight), [1] join, [2] triangulate, [3] hypotheses+score (all rounds), [4] mask+refit,
 * [5] whole call wall (host clock), [6] host replay + RNG (host clock), [7] match kernel launches */
int slam_track_last_timing(double *h_ms8);

#ifdef __cplusplus
}
#endif
#endif

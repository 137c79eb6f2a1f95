"""Drop-in for the hot-path functions of the reference's `utils/utills.py`, running on the MI355X.

Same names, positional signatures, return types and error behaviour as the reference (line numbers below refer to
/root/reference/utils/utills.py); the arithmetic runs in the HIP kernels behind libslam_amd.so.  What stays on the
host is container glue the reference also does in Python (dicts, object arrays, list filtering) and the control flow
of the adaptive RANSAC loop, which is replayed over batched device results.

Differences a caller can observe, all forced by the environment:
  * detection (cv2 AKAZE, :46-54) is out of scope: frames come from a detection source installed with
    `set_detection_source` (anything with `get(frame_num) -> (xy_left, dsc_left, xy_right, dsc_right)`);
  * `cv2.DMatch` / `cv2.KeyPoint` are replaced by the duck types of slam_amd.types;
  * the whole-movie and loop-closure drivers batch all frame pairs into one device pass, so they give every pair its
    own MT19937 stream (seeds drawn once from the global np.random state) instead of threading one global stream
    through the pairs sequentially; the per-pair functions consume the global stream exactly like the reference;
  * `match()` reads MATCH_METHOD when it is called, and the batched drivers read MATCH_METHOD / RATIO /
    CROSSCHECK_RULE the same way, so assigning `utills.MATCH_METHOD = REGULAR_KNN` switches both APIs together (the
    reference binds the default at import, :105, so there the assignment changes nothing);
  * CROSSCHECK_RULE selects which of cv2.BFMatcher(crossCheck=True)'s two historical behaviours is followed.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib, ops, pipeline, tracks, trajectory
from .config import BRUTE_FORCE, FLANN_KNN, MUTUAL, NORM_HAMMING, NORM_HAMMING2, OCV_LEGACY, REGULAR_KNN, HotPathConfig
from .synth import K_KITTI, M1_KITTI, M2_KITTI
from .types import (ArrayDataBase, DataBase, DMatch, DMatchArray, Feature, FeatureArray, Frame, KeyPoint, KeyPointArray,
                    LazyPairFeatures, Track, dmatches_from_arrays, keypoints_from_xy, keypoints_to_xy)

# == module constants, as utills.py:17-42 ==
K, M1, M2 = K_KITTI.copy(), M1_KITTI.copy(), M2_KITTI.copy()
RATIO = 0.6
MATCHES_NORM = NORM_HAMMING2
MATCH_METHOD = BRUTE_FORCE
# which of cv2.BFMatcher(crossCheck=True)'s two historical behaviours the cross-check matcher follows (the reference
# pins no OpenCV version; config.py / DESIGN.md section 2): OCV_LEGACY (default) or MUTUAL
CROSSCHECK_RULE = OCV_LEGACY
PNP_NUM_PTS = 4
SUPP_ERR = 2
REC_ERR = 2
MOVIE_LEN = 3450
IMAGE_HEIGHT, IMAGE_WIDTH = 376, 1241
SOLVEPNP_ITERATIVE, SOLVEPNP_AP3P = 0, 5        # cv2 flag values (:503, :563)
MAX_DRAWS = 64000                               # bound on the reference's unbounded retry loop (:506-507)

DETECTIONS = None


def set_detection_source(src):
    """Installs the object that stands in for KITTI.get_image + ALG.detectAndCompute (:52-53, :393)."""
    global DETECTIONS
    DETECTIONS = src


class ArrayDetections:
    """Detection source over stacked arrays: desc uint8[F,2,n_max,>=61], xy float32[F,2,n_max,2], count int32[F,2]."""

    def __init__(self, desc, xy, count):
        self.desc, self.xy, self.count = desc, xy, count

    def __len__(self):
        return len(self.count)

    def get(self, i):
        n0, n1 = self.count[i]
        return self.xy[i, 0, :n0], self.desc[i, 0, :n0, :61], self.xy[i, 1, :n1], self.desc[i, 1, :n1, :61]

    def arrays(self, frames):
        frames = np.asarray(frames)
        if len(frames) and np.array_equal(frames, np.arange(frames[0], frames[0] + len(frames))):
            frames = slice(int(frames[0]), int(frames[0]) + len(frames))       # a run of frames: views, no 1 GB copy
        desc = self.desc[frames]
        if desc.shape[3] != 64:
            d = np.zeros(desc.shape[:3] + (64,), np.uint8)
            d[..., :desc.shape[3]] = desc
            desc = d
        return desc, np.ascontiguousarray(self.xy[frames]), np.ascontiguousarray(self.count[frames])


def _cuda(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(ops.device())


# == Ex1: matching ==
def _match_arrays(img1_dsc, img2_dsc, norm, mode):
    q = _cuda(ops.pad_descriptors(img1_dsc))
    t = _cuda(ops.pad_descriptors(img2_dsc))
    k1, k2 = ops.match_keys(q, t, norm, mode)
    i1, d1 = (x.cpu().numpy() for x in ops.unpack_keys(k1))
    if k2 is None:
        return i1, d1, None, None
    i2, d2 = (x.cpu().numpy() for x in ops.unpack_keys(k2))
    return i1, d1, i2, d2


class DescriptorArray(np.ndarray):
    """The descriptor matrix read_and_rec_match returns (np.array(dsc), utills.py:54): an ordinary uint8[n, d] array
    that remembers the padded copy of itself already on the device (`_dev`), so the temporal match of the next call
    does not upload it again."""

    def __new__(cls, dsc, dev=None):
        a = np.asarray(dsc).view(cls)
        a._dev = dev
        return a

    def __array_finalize__(self, parent):
        self._dev = None                      # slices, copies: a different matrix


def _dev_descriptors(dsc):
    dev = getattr(dsc, "_dev", None)
    return dev if dev is not None else _cuda(ops.pad_descriptors(dsc))


def _match_qtd(img1_dsc, img2_dsc, method=None):
    """match() (:105-119) as arrays: (queryIdx, trainIdx, distance) of the matches in query order, under MATCHES_NORM and
    the method's rule (cross-check, or kNN + ratio test in float64 as significance_test evaluates it)."""
    method = MATCH_METHOD if method is None else method
    nq = len(img1_dsc)
    if method not in (BRUTE_FORCE, REGULAR_KNN) or nq == 0 or len(img2_dsc) == 0:
        if method == FLANN_KNN:
            knn_flann_matching(img1_dsc, img2_dsc)
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float64)
    q, t = _dev_descriptors(img1_dsc), _dev_descriptors(img2_dsc)
    if method == BRUTE_FORCE:
        mode = _lib.SLAM_MATCH_CROSSCHECK_MUTUAL if CROSSCHECK_RULE == MUTUAL else _lib.SLAM_MATCH_CROSSCHECK
        k1, _ = ops.match_keys(q, t, MATCHES_NORM, mode)
        k1 = k1.cpu().numpy().view(np.uint32)
        keep = np.nonzero(k1 != _lib.SLAM_KEY_NONE)[0]
    else:
        k1, k2 = ops.match_keys(q, t, MATCHES_NORM, _lib.SLAM_MATCH_KNN2)
        k12 = torch.stack([k1, k2]).cpu().numpy().view(np.uint32)
        k1, k2 = k12[0], k12[1]
        both = (k1 != _lib.SLAM_KEY_NONE) & (k2 != _lib.SLAM_KEY_NONE)        # significance_test unpacks two neighbours
        if not both.all() and (k1 != _lib.SLAM_KEY_NONE).any():
            raise ValueError("not enough values to unpack (expected 2, got 1)")       # a train image with one row (:137)
        d1, d2 = (k1 >> _lib.SLAM_KEY_SHIFT).astype(np.float64), (k2 >> _lib.SLAM_KEY_SHIFT).astype(np.float64)
        keep = np.nonzero(both & (d1 < RATIO * d2))[0]
    k = k1[keep]
    return keep.astype(np.int64), (k & ops.IDX_MASK).astype(np.int64), (k >> _lib.SLAM_KEY_SHIFT).astype(np.float64)


def bf_matching(metric, img1_dsc, img2_dsc, crossCheck=True, sort=True):
    """:57-70.  cv2.BFMatcher(metric, crossCheck).match: list of DMatch in query order, or sorted by distance."""
    if crossCheck:
        mode = _lib.SLAM_MATCH_CROSSCHECK_MUTUAL if CROSSCHECK_RULE == MUTUAL else _lib.SLAM_MATCH_CROSSCHECK
        idx, dist, _, _ = _match_arrays(img1_dsc, img2_dsc, metric, mode)
    else:
        if len(img2_dsc) == 0:
            return []
        idx, dist, _, _ = _match_arrays(img1_dsc, img2_dsc, metric, _lib.SLAM_MATCH_KNN2)
    q = np.nonzero(idx >= 0)[0]
    matches = list(dmatches_from_arrays(q, idx[q], dist[q]))
    if sort:
        matches = sorted(matches, key=lambda x: x.distance)
    return matches


def knn_matching(metric, img1_dsc, img2_dsc):
    """:92-102.  cv2.BFMatcher(metric).knnMatch(d1, d2, 2): per query the two nearest train rows."""
    i1, d1, i2, d2 = _match_arrays(img1_dsc, img2_dsc, metric, _lib.SLAM_MATCH_KNN2)
    out = []
    for q in range(len(i1)):
        row = [DMatch(q, i1[q], d1[q])] if i1[q] >= 0 else []
        if i2[q] >= 0:
            row.append(DMatch(q, i2[q], d2[q]))
        out.append(row)
    return out


def knn_flann_matching(img1_dsc, img2_dsc):
    """:73-89 uses an approximate, randomised LSH index (cv2.FlannBasedMatcher): out of scope."""
    raise NotImplementedError("FLANN-LSH matching is approximate and non-deterministic; use BRUTE_FORCE or REGULAR_KNN")


def significance_test(matches, ratio):
    """:131-152."""
    pass_test, failed_test = [], []
    for first, second in matches:
        (pass_test if first.distance < ratio * second.distance else failed_test).append(first)
    return pass_test, failed_test


def match(img1_dsc, img2_dsc, method=None):
    """:105-119.  Default (method = MATCH_METHOD): cross-check brute force under MATCHES_NORM, unsorted, as a numpy
    object array."""
    method = MATCH_METHOD if method is None else method
    if method == BRUTE_FORCE:
        return np.array(bf_matching(MATCHES_NORM, img1_dsc, img2_dsc, sort=False), dtype=object)
    if method == REGULAR_KNN:
        matches = knn_matching(MATCHES_NORM, img1_dsc, img2_dsc)
    elif method == FLANN_KNN:
        matches = knn_flann_matching(img1_dsc, img2_dsc)
    else:
        matches = []
    matches, _ = significance_test(matches, RATIO)
    return np.array(matches, dtype=object)


def detect_and_match(img1, img2):
    """:122-128 with detection replaced by the detection source: img = (xy, dsc).  The key-point and match arrays are
    numpy object arrays, as the reference's, whose KeyPoint / DMatch elements are made when they are read
    (types.LazyObjectArray); the functions below work on the coordinate / index arrays underneath."""
    (xy1, dsc1), (xy2, dsc2) = img1, img2
    d1 = DescriptorArray(dsc1, _cuda(ops.pad_descriptors(dsc1)))
    d2 = DescriptorArray(dsc2, _cuda(ops.pad_descriptors(dsc2)))
    return KeyPointArray(xy1), d1, KeyPointArray(xy2), d2, DMatchArray(*_match_qtd(d1, d2))


# == Ex2: stereo-row rejection, triangulation ==
def axes_of_matches(match, img1_kpts, img2_kpts):
    """:157-165."""
    x1, y1 = img1_kpts[match.queryIdx].pt
    x2, y2 = img2_kpts[match.trainIdx].pt
    return x1, y1, x2, y2


def rectified_stereo_pattern_rej(img1_kpts, img2_kpts, matches):
    """:168-183.  |y_right - y_left| <= REC_ERR, inclusive; returns index lists into `matches`."""
    if isinstance(matches, DMatchArray) and isinstance(img1_kpts, KeyPointArray) and isinstance(img2_kpts, KeyPointArray):
        # float32 coordinates widened to float64, as KeyPoint.pt hands them out
        ok = np.abs(img2_kpts.xy[matches.t, 1].astype(np.float64) - img1_kpts.xy[matches.q, 1].astype(np.float64)) <= REC_ERR
        return np.nonzero(ok)[0].tolist(), np.nonzero(~ok)[0].tolist()
    inl, outl = [], []
    for i in range(len(matches)):
        _, y1, _, y2 = axes_of_matches(matches[i], img1_kpts, img2_kpts)
        (inl if abs(y2 - y1) <= REC_ERR else outl).append(i)
    return inl, outl


def get_matches_coor(matches, img1_kpts, img2_kpts):
    """:186-196."""
    a = [img1_kpts[m.queryIdx].pt for m in matches]
    b = [img2_kpts[m.trainIdx].pt for m in matches]
    return np.array(a), np.array(b)


def triangulate(l_mat, r_mat, kp1_xy_lst, kp2_xy_lst):
    """:242-255.  float64[n, 3] world points by linear (DLT) triangulation."""
    a = np.asarray(kp1_xy_lst, dtype=np.float64).reshape(-1, 2)
    b = np.asarray(kp2_xy_lst, dtype=np.float64).reshape(-1, 2)
    if len(a) == 0:
        return np.array([])
    return ops.triangulate_f64(l_mat, r_mat, _cuda(a), _cuda(b)).cpu().numpy()


def linear_least_square(l_cam_mat, r_cam_mat, kp1_xy, kp2_xy):
    """:220-239.  Unit homogeneous solution of the 4x4 DLT system (sign is arbitrary, as with an SVD)."""
    X = triangulate(l_cam_mat, r_cam_mat, [kp1_xy], [kp2_xy])[0]
    v = np.append(X, 1.0)
    return v / np.linalg.norm(v)


# == Ex3: 4-view join, RANSAC ==
def create_rec_dic(pair_matches, pair_rec_matches_idx):
    """:423-435."""
    return {pair_matches[idx].queryIdx: idx for idx in pair_rec_matches_idx}


def find_kpts_in_all_4_rec(left0_left1_matches, rec0_dic, rec1_dic):
    """:438-471."""
    q0, q1, q01 = [], [], []
    for i, m in enumerate(left0_left1_matches):
        if m.queryIdx in rec0_dic and m.trainIdx in rec1_dic:
            q0.append(rec0_dic[m.queryIdx])
            q1.append(rec1_dic[m.trainIdx])
            q01.append(i)
    return q0, q1, q01


def tracking_4images(left0_dsc, left1_dsc, pair0_matches, pair0_rec_matches_idx, pair1_matches, pair1_rec_matches_idx):
    """:945-960."""
    if isinstance(pair0_matches, DMatchArray) and isinstance(pair1_matches, DMatchArray):
        # the same dictionaries and scan as below, on the index arrays (a later duplicate key overwrites, as in a dict)
        tq, tt, _ = _match_qtd(left0_dsc, left1_dsc)
        r0, r1 = np.asarray(pair0_rec_matches_idx, dtype=np.int64), np.asarray(pair1_rec_matches_idx, dtype=np.int64)
        tab0 = np.full(max(len(left0_dsc), 1), -1, np.int64)
        tab1 = np.full(max(len(left1_dsc), 1), -1, np.int64)
        tab0[pair0_matches.q[r0]] = r0
        tab1[pair1_matches.q[r1]] = r1
        i0, i1 = tab0[tq], tab1[tt]
        keep = (i0 >= 0) & (i1 >= 0)
        return pair0_matches[i0[keep]], pair1_matches[i1[keep]]
    left0_left1_matches = match(left0_dsc, left1_dsc)
    rec0 = create_rec_dic(pair0_matches, pair0_rec_matches_idx)
    rec1 = create_rec_dic(pair1_matches, pair1_rec_matches_idx)
    i0, i1, _ = find_kpts_in_all_4_rec(left0_left1_matches, rec0, rec1)
    return np.asarray(pair0_matches, dtype=object)[i0], np.asarray(pair1_matches, dtype=object)[i1]


def compose_transformations(first_ex_mat, second_ex_mat):
    """:659-669."""
    hom1 = np.append(first_ex_mat, [np.array([0, 0, 0, 1])], axis=0)
    return second_ex_mat @ hom1


def project(p3d_pts, cam_proj_mat):
    """:698-705 (host helper; the hot path projects inside the scoring kernel)."""
    hom = p3d_pts @ cam_proj_mat[:, :3].T + cam_proj_mat[:, 3].T
    return hom[:, :2] / hom[:, [-1]]


def compute_square_dist(pts_lst1, pts_lst2=None, dim="3d"):
    """:719-736, including the quirk that the default returns the UN-squared norm."""
    if pts_lst2 is None:
        pts_lst2 = np.zeros(pts_lst1.shape)
    sub = pts_lst1 - pts_lst2
    if dim == "2d":
        return np.einsum("ij,ij->i", sub, sub)
    if dim == "3d":
        return np.linalg.norm(sub, axis=1)
    return 0


def check_projection(img_projected_pts, img_pts_coor, acc=SUPP_ERR):
    """:708-716."""
    return compute_square_dist(img_projected_pts, img_pts_coor) <= acc ** 2


def _as_f64_points(world, left, right):
    X = np.ascontiguousarray(np.asarray(world, dtype=np.float64).reshape(-1, 3))
    l = np.ascontiguousarray(np.asarray(left, dtype=np.float64).reshape(-1, 2))
    r = np.ascontiguousarray(np.asarray(right, dtype=np.float64).reshape(-1, 2))
    return X, l, r


def consensus_match(calib_mat, left1_ex_mat, right1_to_left1_ex_mat, world_p3d_pts, left1_matches_coor,
                    right1_matches_coor, acc):
    """:533-545.  bool[n]: projection within acc**2 px (un-squared distance) in left1 AND right1."""
    X, l, r = _as_f64_points(world_p3d_pts, left1_matches_coor, right1_matches_coor)
    if len(X) == 0:
        return np.zeros(0, bool)
    m = ops.consensus_match_f64(calib_mat, right1_to_left1_ex_mat, left1_ex_mat, _cuda(X), _cuda(l), _cuda(r), acc ** 2)
    return m.cpu().numpy().astype(bool)


def find_supporters_4_frame(pair0_p3d_pts, left1_proj_mat, left1_matches_coor, right1_proj_mat, right1_matches_coor,
                            acc=SUPP_ERR):
    """:672-695 (host helper on ready-made projection matrices)."""
    return check_projection(project(pair0_p3d_pts, left1_proj_mat), left1_matches_coor, acc) & \
        check_projection(project(pair0_p3d_pts, right1_proj_mat), right1_matches_coor, acc)


class _Device4View:
    """Points of one frame pair on the device in the batched layout (one segment)."""

    def __init__(self, world, left1, right1):
        X, l, r = _as_f64_points(world, left1, right1)
        self.n = len(X)
        self.stride = max(64, (self.n + 63) // 64 * 64)
        Xs = np.zeros((1, self.stride, 3))
        Xs[0, :self.n] = X
        xy = np.zeros((1, self.stride, 4))
        xy[0, :self.n, :2] = l
        xy[0, :self.n, 2:] = r
        self.X, self.xy = _cuda(Xs), _cuda(xy)
        self.count = _cuda(np.array([self.n], np.int32))

    @classmethod
    def on_device(cls, X, xy, count, n):
        """X float64[1, stride, 3], xy float32 | float64 [1, stride, 4], count int32[1] already on the device."""
        v = cls.__new__(cls)
        v.n, v.stride, v.X, v.xy, v.count = int(n), int(X.shape[1]), X, xy, count
        return v


def _iter_estimate(prob, outliers_perc, model_parm_num):
    """:488-489."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log(1 - prob) / np.log(1 - np.power(1 - outliers_perc, model_parm_num))


def _ransac_global_stream(pts, right1_to_left1_ex_mat, calib_mat, thr, model_parm_num=PNP_NUM_PTS):
    """The loop of :498-523 + :526-530 over points already on the device, drawing from the GLOBAL np.random stream in
    chunks; the loop's bookkeeping is replayed over each chunk by the library (slam_ransac_replay, the code the shard
    pipeline uses) and the stream is rewound to exactly the draws the reference would have consumed."""
    n = pts.n
    L = _lib.lib()
    st = _lib.SlamRansacState()
    L.slam_ransac_state_init(ctypes.byref(st))
    state = np.random.get_state()
    prob, max_iterations = 0.9999, 1e3
    best_pose, chunk, begin = None, 64, 0
    while not st.finished:
        sample = np.random.randint(0, n, (chunk, model_parm_num)).astype(np.int32)
        pose, status = ops.p3p_hypotheses(calib_mat, pts.X, pts.xy, pts.count, _cuda(sample[None]))
        cnt, _ = ops.score_hypotheses(calib_mat, right1_to_left1_ex_mat, pts.X, pts.xy, pts.count, pose, status, thr)
        h_status, h_cnt = np.ascontiguousarray(status.cpu().numpy()[0]), np.ascontiguousarray(cnt.cpu().numpy()[0])
        before = st.best_draw
        _lib.check(L.slam_ransac_replay(ctypes.byref(st), n, h_status.ctypes.data, h_cnt.ctypes.data, begin, begin + chunk,
                                        prob, max_iterations, 0, MAX_DRAWS))
        if st.hit_cap:
            raise RuntimeError(f"no PnP-RANSAC termination within {MAX_DRAWS} draws (the reference would not return)")
        if st.best_draw != before:
            best_pose = pose[0, st.best_draw - begin].clone()
        begin += chunk
        chunk = min(chunk * 4, 4096)
    np.random.set_state(state)                                   # leave the stream where the reference would
    if st.draws:
        np.random.randint(0, n, (st.draws, model_parm_num))
    if best_pose is None:
        raise TypeError("'NoneType' object is not iterable")     # sum(None) in extra_refine (:558)
    one = torch.ones((1, 1), dtype=torch.uint8, device=pts.X.device)
    _, words = ops.score_hypotheses(calib_mat, right1_to_left1_ex_mat, pts.X, pts.xy, pts.count, best_pose[None, None],
                                    one, thr, want_mask=True)
    T, status = ops.refit_pnp(calib_mat, pts.X, pts.xy, pts.count, words[0], best_pose[None])
    mask = ops.mask_words_to_bool(words.cpu().numpy()[0, 0], n)
    return (T.cpu().numpy()[0] if int(status[0]) else None), mask


def online_est_pnp_ransac(model_parm_num, world_p3d_pts, left0_ex_mat, left0_matches_coor, right0_ex_mat,
                          right0_matches_coor, left1_matches_coor, right1_to_left1_ex_mat, right1_matches_coor,
                          calib_mat=K, acc=SUPP_ERR):
    """:474-530.  Adaptive PnP-RANSAC; returns (refit pose | None, PRE-refit best supporter mask).

    Draws come from the global np.random stream exactly as in the reference (4 indices per draw, failed solves
    consume a draw without counting as an iteration).  Hypotheses are generated and scored on the device in chunks;
    the stopping rule is replayed over the chunk and the stream is rewound to what the reference would have used."""
    if model_parm_num != PNP_NUM_PTS:
        raise NotImplementedError("the minimal solver is P4P/AP3P (PNP_NUM_PTS = 4)")
    n = len(left1_matches_coor)
    if n == 0:
        np.random.randint(0, 0, model_parm_num)                 # raises ValueError like the reference (:502)
    pts = _Device4View(world_p3d_pts, left1_matches_coor, right1_matches_coor)
    return _ransac_global_stream(pts, right1_to_left1_ex_mat, calib_mat, acc ** 2, model_parm_num)


def _dlt_pose(world, img, calib_mat):
    """The start cv2.solvePnP(flags=SOLVEPNP_ITERATIVE) gives its Levenberg-Marquardt solver for a non-planar set
    (cvFindExtrinsicCameraParams2, called at :592 through :562-563): the 12 entries of [R|t] as the null vector of the
    2n x 12 DLT system on K-normalised image points, the rotation part projected onto SO(3) (U V^T of its SVD) and the
    translation rescaled by the same factor.  A 12 x 12 eigenproblem: host numpy, like the reference's own glue."""
    fx, fy, cx, cy = calib_mat[0][0], calib_mat[1][1], calib_mat[0][2], calib_mat[1][2]
    n = len(world)
    mx, my = (img[:, 0] - cx) / fx, (img[:, 1] - cy) / fy
    L = np.zeros((2 * n, 12))
    L[0::2, 0:3], L[0::2, 3] = world, 1.0
    L[0::2, 8:11], L[0::2, 11] = -mx[:, None] * world, -mx
    L[1::2, 4:7], L[1::2, 7] = world, 1.0
    L[1::2, 8:11], L[1::2, 11] = -my[:, None] * world, -my
    _, _, Vt = np.linalg.svd(L.T @ L)
    RRt = Vt[11].reshape(3, 4)
    if np.linalg.det(RRt[:, :3]) < 0:
        RRt = -RRt
    U, _, Vt2 = np.linalg.svd(RRt[:, :3])
    R = U @ Vt2
    t = RRt[:, 3] * (np.linalg.norm(R) / np.linalg.norm(RRt[:, :3]))
    return np.hstack((R, t.reshape(3, 1)))


def pnp(world_p3d_pts, img_proj_coor, calib_mat, flag, init=None):
    """:585-596.  [R|t] 3x4 or None.  flag = SOLVEPNP_AP3P (exactly 4 points; None when AP3P has no root) or
    SOLVEPNP_ITERATIVE (>= 6 points: the DLT start of cv2's solver, _dlt_pose, or `init` when the caller has a pose --
    the RANSAC winner in the hot path -- then the Levenberg-Marquardt refit on the device; like cv2 it returns a pose
    for any >= 6 points, and None below that as the caller's guard at :558-559 expects)."""
    world = np.asarray(world_p3d_pts, dtype=np.float64).reshape(-1, 3)
    img = np.asarray(img_proj_coor, dtype=np.float64).reshape(-1, 2)
    pts = _Device4View(world, img, img)
    n = pts.n
    if flag == SOLVEPNP_AP3P:
        if n != 4:
            raise ValueError("AP3P needs exactly 4 points")
        pose, status = ops.p3p_hypotheses(calib_mat, pts.X, pts.xy, pts.count, _cuda(np.arange(4, dtype=np.int32)[None, None]))
        return pose.cpu().numpy()[0, 0] if int(status[0, 0]) else None
    if n < 6:
        return None
    if init is None:
        with np.errstate(all="ignore"):
            init = _dlt_pose(world, img, np.asarray(calib_mat, dtype=np.float64))
    init_t = _cuda(np.asarray(init, dtype=np.float64).reshape(1, 3, 4))
    words = _cuda(ops.bool_to_mask_words(np.ones((1, n), bool), pts.stride))
    T, status = ops.refit_pnp(calib_mat, pts.X, pts.xy, pts.count, words, init_t)
    return T.cpu().numpy()[0] if int(status[0]) else None


def extra_refine(supp_indicators, world_p3d_pts, left1_matches_coor, right1_to_left1_ex_mat, right1_matches_coor,
                 calib_mat=K, acc=SUPP_ERR, it_num=1, init=None):
    """:548-571."""
    supp = np.asarray(supp_indicators, dtype=bool)
    left1_ex_mat, new_mask = None, [False] * len(supp)
    if supp.sum() < 6:
        return None, new_mask
    world, left = np.asarray(world_p3d_pts), np.asarray(left1_matches_coor)
    for _ in range(it_num):
        left1_ex_mat = pnp(world[supp], left[supp], calib_mat, SOLVEPNP_ITERATIVE, init=init)
        if left1_ex_mat is None:
            continue
        new_mask = consensus_match(calib_mat, left1_ex_mat, right1_to_left1_ex_mat, world_p3d_pts, left1_matches_coor,
                                   right1_matches_coor, acc)
    return left1_ex_mat, new_mask


# == poses / trajectory ==
def relative_camera_pos(ex_cam_mat):
    """:599-606."""
    return -1 * ex_cam_mat[:, :3].T @ ex_cam_mat[:, 3]


def convert_ex_cam_to_cam_to_world(ex_cam):
    """:1025-1034."""
    R, t = ex_cam[:, :3], ex_cam[:, 3]
    return np.hstack((R.T, (-R.T @ t).reshape(3, 1)))


def between(first_cam, second_cam):
    """:653-656."""
    return compose_transformations(convert_ex_cam_to_cam_to_world(first_cam), second_cam)


def convert_trans_from_rel_to_global(T_arr):
    """:749-757."""
    out, last = [], T_arr[0]
    for t in T_arr:
        last = compose_transformations(last, t)
        out.append(last)
    return out


def left_cameras_trajectory(relative_T_arr):
    """:785-794."""
    return np.array([relative_camera_pos(t) for t in relative_T_arr])


def cameras_rel_pnp_est_and_ground_truth_locations(db, ground_truth=None):
    """:1200-1208.  Camera-centre offsets between the PnP trajectory and the ground truth (the trajectory-error
    series the reference plots); `ground_truth` replaces the KITTI poses file (:771-782)."""
    est = convert_trans_from_rel_to_global(db.get_relative_cam_trans())
    if ground_truth is None:
        raise ValueError("pass ground_truth: the KITTI poses file is not available")
    return left_cameras_trajectory([between(a, b) for a, b in zip(est, ground_truth)])


def rot_mat_to_euler_angles(R_mat):
    """:1072-1086."""
    return trajectory.euler_angles(R_mat)


def camera_angles(ex_cam_mat):
    """:631-638."""
    return rot_mat_to_euler_angles(ex_cam_mat[:, :3])


def left_cameras_angles(global_trans):
    """:821-830."""
    return np.array([camera_angles(t) for t in global_trans])


def translation_to_vec(translation):
    """:1089-1094.  (euler angles, camera centre) as one 6-vector."""
    R_mat, t_vec = translation[:, :3], translation[:, 3]
    return np.hstack((rot_mat_to_euler_angles(R_mat), -R_mat.T @ t_vec))


def cameras_rel_pnp_est_and_ground_truth_angles(db, ground_truth=None):
    """:1211-1222.  Euler angles of R_est^T R_gt per frame; `ground_truth` replaces the KITTI poses file."""
    if ground_truth is None:
        raise ValueError("pass ground_truth: the KITTI poses file is not available")
    est = convert_trans_from_rel_to_global(db.get_relative_cam_trans())
    return trajectory.angle_errors(est, ground_truth)


def trajectory_error_summary(db, ground_truth, delta=1):
    """ATE / RPE figures of the PnP trajectory (no counterpart in the reference, which only plots the series above)."""
    return trajectory.summary(convert_trans_from_rel_to_global(db.get_relative_cam_trans()), ground_truth, delta)


# == Ex4: per-pair and whole-movie drivers ==
def get_feature_obj(matches, img1_kpts, img2_kpts):
    """:963-975."""
    if isinstance(matches, DMatchArray) and isinstance(img1_kpts, KeyPointArray) and isinstance(img2_kpts, KeyPointArray):
        return FeatureArray(np.stack([matches.q, matches.t], 1), np.hstack([img1_kpts.xy[matches.q], img2_kpts.xy[matches.t]]))
    out = np.empty(len(matches), dtype=object)
    for i, m in enumerate(matches):
        (x1, y1), (x2, y2) = img1_kpts[m.queryIdx].pt, img2_kpts[m.trainIdx].pt
        out[i] = Feature(m.queryIdx, m.trainIdx, x1, x2, y1, y2)
    return out


def get_features_left_coor(frame_features):
    """:978-986."""
    if isinstance(frame_features, FeatureArray):
        return frame_features.coor[:, 0:2].astype(np.float64) if len(frame_features) else np.array([])
    return np.array([f.get_left_coor() for f in frame_features])


def get_features_right_coor(frame_features):
    """:989-997."""
    if isinstance(frame_features, FeatureArray):
        return frame_features.coor[:, 2:4].astype(np.float64) if len(frame_features) else np.array([])
    return np.array([f.get_right_coor() for f in frame_features])


def read_and_rec_match(frame_num):
    """:386-396 with detection served by the detection source."""
    if DETECTIONS is None:
        raise RuntimeError("no detection source: call slam_amd.utills.set_detection_source(...) first")
    xy_l, dsc_l, xy_r, dsc_r = DETECTIONS.get(frame_num)
    l_kpts, l_dsc, r_kpts, _, matches = detect_and_match((xy_l, dsc_l), (xy_r, dsc_r))
    rec_idx, _ = rectified_stereo_pattern_rej(l_kpts, r_kpts, matches)
    return l_kpts, l_dsc, r_kpts, matches, rec_idx


def find_features_in_consecutive_frames(left0_kpts, left0_dsc, right0_kpts, pair0_matches, pair0_rec_matches_idx,
                                        left1_kpts, left1_dsc, right1_kpts, pair1_matches, pair1_rec_matches_idx):
    """:876-925.  -> (pose, frame0 features, frame1 features, inlier %, supporter mask)."""
    if all(isinstance(k, KeyPointArray) for k in (left0_kpts, right0_kpts, left1_kpts, right1_kpts)) and \
            isinstance(pair0_matches, DMatchArray) and isinstance(pair1_matches, DMatchArray):
        return _consecutive_frames_on_arrays(left0_kpts, left0_dsc, right0_kpts, pair0_matches, pair0_rec_matches_idx,
                                             left1_kpts, left1_dsc, right1_kpts, pair1_matches, pair1_rec_matches_idx)
    t0, t1 = tracking_4images(left0_dsc, left1_dsc, pair0_matches, pair0_rec_matches_idx, pair1_matches,
                              pair1_rec_matches_idx)
    frame0_features = get_feature_obj(t0, left0_kpts, right0_kpts)
    frame1_features = get_feature_obj(t1, left1_kpts, right1_kpts)
    left0, right0 = get_features_left_coor(frame0_features), get_features_right_coor(frame0_features)
    X = triangulate(K @ M1, K @ M2, left0, right0)
    left1, right1 = get_features_left_coor(frame1_features), get_features_right_coor(frame1_features)
    trans, mask = online_est_pnp_ransac(PNP_NUM_PTS, X, M1, left0, M2, right0, left1, M2, right1, K, acc=SUPP_ERR)
    perc = 100 * sum(mask) / len(mask)
    return trans, frame0_features, frame1_features, perc, mask


def _consecutive_frames_on_arrays(left0_kpts, left0_dsc, right0_kpts, pair0_matches, pair0_rec_matches_idx,
                                  left1_kpts, left1_dsc, right1_kpts, pair1_matches, pair1_rec_matches_idx):
    """find_features_in_consecutive_frames (:876-925) when the key-point / match arrays are the lazily materialised ones
    read_and_rec_match hands out: the same steps in the same order (:890-923), on the coordinate / index arrays
    underneath, with the triangulated points staying on the device between triangulation and the RANSAC loop."""
    t0, t1 = tracking_4images(left0_dsc, left1_dsc, pair0_matches, pair0_rec_matches_idx, pair1_matches,
                              pair1_rec_matches_idx)
    frame0_features = get_feature_obj(t0, left0_kpts, right0_kpts)
    frame1_features = get_feature_obj(t1, left1_kpts, right1_kpts)
    n = len(frame0_features)
    if n == 0:
        np.random.randint(0, 0, PNP_NUM_PTS)                    # raises ValueError like the reference (:502)
    stride = max(64, (n + 63) // 64 * 64)
    xy = np.zeros((2, stride, 4), np.float32)                   # segment 0: frame 0 (for triangulation), segment 1: frame 1
    xy[0, :n], xy[1, :n] = frame0_features.coor, frame1_features.coor
    d_xy = _cuda(xy)
    count = _cuda(np.array([n], np.int32))
    X = ops.triangulate(K @ M1, K @ M2, d_xy[0:1], count)
    pts = _Device4View.on_device(X, d_xy[1:2], count, n)
    trans, mask = _ransac_global_stream(pts, M2, K, SUPP_ERR ** 2)
    perc = 100 * mask.sum() / len(mask)                         # :923; numpy int / int, as sum() of a bool array gives
    return trans, frame0_features, frame1_features, perc, mask


def compute_trans_between_cur_to_next(left0_kpts, left0_dsc, right0_kpts, pair0_matches, pair0_rec_matches_idx,
                                      left1_kpts, left1_dsc, right1_kpts, pair1_matches, pair1_rec_matches_idx):
    """:259-293.  -> (pose, supporter mask)."""
    trans, _, _, _, mask = find_features_in_consecutive_frames(left0_kpts, left0_dsc, right0_kpts, pair0_matches,
                                                               pair0_rec_matches_idx, left1_kpts, left1_dsc, right1_kpts,
                                                               pair1_matches, pair1_rec_matches_idx)
    return trans, mask


def _hot_path_config() -> HotPathConfig:
    """The module constants as the batched drivers' configuration: the whole-movie and loop-closure drivers match the
    way match() does (MATCH_METHOD, RATIO, CROSSCHECK_RULE), as in the reference, where both go through match()."""
    return HotPathConfig(K=K, M1=M1, M2=M2, norm=MATCHES_NORM, rec_err=float(REC_ERR), supp_err=float(SUPP_ERR),
                         match_method=MATCH_METHOD, ratio=float(RATIO), crosscheck_rule=CROSSCHECK_RULE)


def _source_arrays(frames):
    if DETECTIONS is None:
        raise RuntimeError("no detection source: call slam_amd.utills.set_detection_source(...) first")
    if hasattr(DETECTIONS, "arrays"):
        return DETECTIONS.arrays(frames)
    per = [DETECTIONS.get(int(f)) for f in frames]
    n_max = (max(max(len(p[0]), len(p[2])) for p in per) + 63) // 64 * 64
    desc = np.zeros((len(per), 2, n_max, 64), np.uint8)
    xy = np.zeros((len(per), 2, n_max, 2), np.float32)
    count = np.zeros((len(per), 2), np.int32)
    for i, (xl, dl, xr, dr) in enumerate(per):
        for s, (x, d) in enumerate(((xl, dl), (xr, dr))):
            count[i, s] = len(x)
            xy[i, s, :len(x)] = x
            desc[i, s, :len(x), :np.shape(d)[1]] = d
    return desc, xy, count


def _fit_keys(stereo_key, n_max):
    """Stereo keys int32[F, w] (any device) at the width of this call's arrays: rows past an image's count hold
    SLAM_KEY_NONE, so cutting or padding with it changes nothing."""
    k = stereo_key.to(ops.device())
    if k.shape[1] > n_max:
        k = k[:, :n_max]
    elif k.shape[1] < n_max:
        wide = torch.full((k.shape[0], n_max), -1, dtype=torch.int32, device=k.device)
        wide[:, :k.shape[1]] = k
        k = wide
    return k.contiguous()


def track_pairs_arrays(frames, pair_f0, pair_f1, seeds=None, host_kp=True, stereo_key=None):
    """Batched core of the drivers below: frames = movie indices, pair_f0/f1 = positions in `frames`.
    Returns (ShardResult, xy float32[F,2,n_max,2], tracked key-point indices int32[P,n_max,4]); with host_kp=False the
    result keeps the supporter masks as bit words and the indices stay on the device (ShardResult.tracked_kp).
    stereo_key: the frames' stereo keys (`stereo_keys_of`) when they were matched earlier, possibly on another rank."""
    desc, xy, count = _source_arrays(frames)
    F, _, n_max, _ = desc.shape
    P = len(pair_f0)
    if seeds is None:
        seeds = np.random.randint(0, 2 ** 32, P, dtype=np.uint64)      # one draw from the global stream per pair
    trk = pipeline.ShardTracker(F, P, n_max, _hot_path_config())
    res = trk.run(_cuda(desc.reshape(2 * F, n_max, 64)), _cuda(xy.reshape(2 * F, n_max, 2)), _cuda(count.reshape(2 * F)),
                  np.asarray(pair_f0, np.int32), np.asarray(pair_f1, np.int32), seeds, unpack_mask=host_kp,
                  stereo_key=_fit_keys(stereo_key, n_max) if stereo_key is not None else None)
    return res, xy, (res.tracked_kp.cpu().numpy() if host_kp else None)


def stereo_keys_of(frames):
    """The stereo match of each of `frames` (movie indices) as packed device keys int32[F, n_max]: what
    read_and_rec_match's match() call (:386-396) produces, once per frame."""
    desc, _, count = _source_arrays(frames)
    F, _, n_max, _ = desc.shape
    return pipeline.stereo_keys(_cuda(desc.reshape(2 * F, n_max, 64)), _cuda(count.reshape(2 * F)), _hot_path_config())


def _features_of(kp_lr, xy_frame):
    """key-point index pairs int[n, 2] (left, right) of one frame -> object array of Feature."""
    out = np.empty(len(kp_lr), dtype=object)
    xl, xr = xy_frame[0], xy_frame[1]
    for i, (a, b) in enumerate(kp_lr):
        out[i] = Feature(int(a), int(b), float(xl[a, 0]), float(xr[b, 0]), float(xl[a, 1]), float(xr[b, 1]))
    return out


def find_features_in_consecutive_frames_whole_movie(first_left_ex_cam_mat=M1, movie_len=None, seeds=None):
    """:834-873.  -> (consecutive_frame_features, inliers_percentage, global_trans, relative_trans): the four
    constructor arguments of DataBase (DataBase.py:39).

    consecutive_frame_features is a LazyPairFeatures: a sequence with one [frame-i Feature array, frame-(i+1) Feature
    array] entry per pair, as the reference's list, whose entries are built from the supporters' index arrays when they
    are read (a 3450-frame movie holds ~4.5 M Features; building them all up front took 8-12 s around a 17 ms device
    pass).  DataBase(...) recognises it and chains the tracks on the index arrays (types.ArrayDataBase)."""
    n = movie_len if movie_len is not None else min(MOVIE_LEN, len(DETECTIONS)) if hasattr(DETECTIONS, "__len__") else MOVIE_LEN
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        pose, stats, flat, ptr, xy = _whole_movie_sharded(n, seeds)
    else:
        f0, f1 = pipeline.consecutive_pairs(n)
        res, xy, _ = track_pairs_arrays(np.arange(n), f0, f1, seeds, host_kp=False)
        pose, stats = res.pose, res.stats
        if not (stats[:, 5] & _lib.SLAM_PAIR_EMPTY).any():
            flat, ptr = ops.compact_supporters(res.mask, stats[:, 0], res.tracked_kp)
    if (stats[:, 5] & _lib.SLAM_PAIR_EMPTY).any():
        raise ValueError("low >= high")          # np.random.randint(0, 0, 4) on an empty tracked set (:502)
    sup = [flat[ptr[p]:ptr[p + 1]] for p in range(n - 1)]
    T = stats[:, 0].astype(np.int64)
    percs = (100 * stats[:, 1].astype(np.int64) / T).tolist()              # :923, true division of Python ints
    ok = (stats[:, 5] & _lib.SLAM_PAIR_OK) != 0
    rel = [first_left_ex_cam_mat] + [pose[p] if ok[p] else None for p in range(n - 1)]
    return LazyPairFeatures(sup, xy, flat=(flat, ptr)), percs, convert_trans_from_rel_to_global(rel), rel


def _whole_movie_sharded(n, seeds):
    """The loop of :850 cut into one contiguous block of frame pairs per rank (one process per GPU, torch.distributed
    initialised by the caller; README.md:58 of the reference: the pairs "can be done in parallel").  Rank r reads and
    tracks only its frames (one halo frame); the per-pair seeds are rank 0's draws from ITS global random stream, so the
    result is what one GPU returns for the same stream.  Exchanges: the seeds (broadcast), poses + counters, the
    supporters' key-point rows and the frames' coordinates (all-gathers; RCCL over xGMI under the nccl backend).  Every
    rank returns the whole movie's arrays."""
    import torch.distributed as tdist
    from . import dist as sdist
    rank, world = tdist.get_rank(), tdist.get_world_size()
    P = n - 1
    if seeds is None and rank == 0:
        seeds = np.random.randint(0, 2 ** 32, P, dtype=np.uint64)      # one draw from the global stream per pair
    seeds = sdist.broadcast_array(seeds if rank == 0 else None, (P,), np.int64).astype(np.uint64)
    a, b, fa, fb = sdist.shard_frames(n, rank, world)
    xy_local = np.zeros((0, 2, 64, 2), np.float32)
    pose_l, stats_l = np.zeros((0, 3, 4)), np.zeros((0, 8), np.int32)
    flat_l, cnt_l = np.zeros((0, 4), np.int32), np.zeros(0, np.int64)
    if b > a:
        f0, f1 = pipeline.consecutive_pairs(fb - fa)
        res, xy_local, _ = track_pairs_arrays(np.arange(fa, fb), f0, f1, seeds[a:b], host_kp=False)
        pose_l, stats_l = res.pose, res.stats
        if (stats_l[:, 5] & _lib.SLAM_PAIR_EMPTY).any():                # raised after the gather, on every rank alike
            cnt_l = np.zeros(b - a, np.int64)
        else:
            flat_l, ptr_l = ops.compact_supporters(res.mask, stats_l[:, 0], res.tracked_kp)
            cnt_l = np.diff(ptr_l)
    pose, stats = sdist.gather_shard_results(pose_l, stats_l, P)
    flat, ptr = sdist.gather_ragged_rows(flat_l, cnt_l, P)
    n_max = sdist.max_over_ranks(xy_local.shape[2])                     # ranks pad their own frames; one width for the movie
    if xy_local.shape[2] != n_max:
        wide = np.zeros(xy_local.shape[:2] + (n_max, 2), np.float32)
        wide[:, :, :xy_local.shape[2]] = xy_local
        xy_local = wide
    xy = sdist.gather_frames(xy_local, n)
    return pose, stats, flat, ptr, xy


def whole_movie_database(first_left_ex_cam_mat=M1, movie_len=None, seeds=None):
    """DataBase(*find_features_in_consecutive_frames_whole_movie(...)) (Data.py:42-43) in one call; an ArrayDataBase."""
    return DataBase(*find_features_in_consecutive_frames_whole_movie(first_left_ex_cam_mat, movie_len, seeds))


def whole_movie(first_left_ex_mat=M1, movie_len=None, seeds=None):
    """:399-420.  Relative poses only."""
    return find_features_in_consecutive_frames_whole_movie(first_left_ex_mat, movie_len, seeds)[3]


# == small host helpers of the reference module, kept so that its callers find them here ==
TRIANGULATION_METERS_TO_PIX_DIFF = {450: 1, 250: 2, 150: 3, 120: 4, 90: 5}        # :35
TRIANGULATION_REJ_METERS = 150                                                      # :36
LEFT_CAM_TRANS_PATH = None      # :28 is a path on the reference author's machine; pass the KITTI poses file explicitly


def get_matches_coor_and_kpts(matches, img1_kpts, img2_kpts):
    """:199-211.  One row [img1 key-point index, x1, y1, x2, y2] per match."""
    rows = []
    for m in matches:
        (x1, y1), (x2, y2) = img1_kpts[m.queryIdx].pt, img2_kpts[m.trainIdx].pt
        rows.append([m.queryIdx, x1, y1, x2, y2])
    return np.array(rows)


def get_inliers_and_outliers_coor_for_rec(inliers_matches, outliers_matches, img1_kpts, img2_kpts):
    """:214-217."""
    return get_matches_coor(inliers_matches, img1_kpts, img2_kpts) + get_matches_coor(outliers_matches, img1_kpts, img2_kpts)


def rodriguez_to_mat(R_vec, t_vec):
    """:574-582.  [R|t] from a rotation vector (cv2.Rodrigues: R = cos(th) I + (1 - cos(th)) k k^T + sin(th) [k]x) and a
    translation column."""
    r = np.asarray(R_vec, dtype=np.float64).reshape(3)
    th = float(np.sqrt(r @ r))
    if th < np.finfo(np.float64).eps:
        R = np.eye(3)
    else:
        k = r / th
        kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
        R = np.cos(th) * np.eye(3) + (1.0 - np.cos(th)) * np.outer(k, k) + np.sin(th) * kx
    return np.hstack((R, np.asarray(t_vec, dtype=np.float64).reshape(3, 1)))


def relative_camera_pos_4(left0_ex_mat, right0_ex_mat, left1_ex_mat, right1_ex_mat):
    """:641-650."""
    return tuple(relative_camera_pos(m) for m in (left0_ex_mat, right0_ex_mat, left1_ex_mat, right1_ex_mat))


def euclidean_dist(pts_lst1, pts_lst2=None, dim="3d"):
    """:739-746 (the square root of compute_square_dist, quirk of its "3d" branch included)."""
    return np.sqrt(compute_square_dist(pts_lst1, pts_lst2, dim=dim))


def get_ground_truth_transformations(left_cam_trans_path=LEFT_CAM_TRANS_PATH, movie_len=MOVIE_LEN):
    """:771-782.  The first movie_len 3x4 matrices of a KITTI poses file (12 numbers per line)."""
    if left_cam_trans_path is None:
        raise FileNotFoundError("pass the path of the KITTI poses file (dataset/poses/00.txt)")
    out = []
    with open(left_cam_trans_path) as fh:
        for _, line in zip(range(movie_len), fh):
            out.append(np.array(line.split(), dtype=float).reshape(3, 4))
    return out


def compute_ground_truth_traj(left_cam_trans_path=LEFT_CAM_TRANS_PATH):
    """:356-359."""
    return left_cameras_trajectory(get_ground_truth_transformations(left_cam_trans_path))


def far_or_neg_pts_rej(left0_matches_coor, right0_matches_coor):
    """:928-937.  Disparity test: points nearer than TRIANGULATION_REJ_METERS and in front of the rig."""
    disparity = left0_matches_coor[:, 0] - right0_matches_coor[:, 0]
    return (disparity > TRIANGULATION_METERS_TO_PIX_DIFF[TRIANGULATION_REJ_METERS]) * (disparity > 0)


def get_rand_track(track_len, tracks):
    """:1000-1012.  Draws from `random` until a track of that length comes up (as the reference: forever if there is none)."""
    import random
    while True:
        t = tracks[random.randint(0, len(tracks) - 1)]
        if t.get_track_len() == track_len:
            return t


def compare_dbs_median_tracks_len(db_path, percentage):
    """:940-942."""
    from . import types
    return types.load(db_path).tracks_median_len(percentage)


def mahalanobis_dist(delta, cov):
    """:1052-1054."""
    return (delta.T @ np.linalg.inv(cov) @ delta) ** 0.5


def compute_whole_movie_time(first_left_ex_mat=M1):
    """:362-370.  -> (relative poses, minutes)."""
    import time
    t0 = time.perf_counter()
    rel = find_features_in_consecutive_frames_whole_movie(first_left_ex_mat)[3]
    return rel, (time.perf_counter() - t0) / 60


# == Ex7: loop-closure candidate validation ==
def create_little_tracks(first_frame_features_obj, second_frame_features_obj, first_frame_id, second_frame_id):
    """:1149-1167."""
    tracks = []
    for i in range(len(first_frame_features_obj)):
        t = Track(i)
        t.add_feature(first_frame_id, first_frame_features_obj[i])
        t.add_feature(second_frame_id, second_frame_features_obj[i])
        tracks.append(t)
    return tracks


def apply_full_consensus_match(first_frame, second_frame):
    """:1103-1115."""
    a, b = read_and_rec_match(first_frame), read_and_rec_match(second_frame)
    trans, f0, f1, perc, mask = find_features_in_consecutive_frames(*a, *b)
    return trans, f0, f1, mask, perc


def _pair_supporter_rows(frames, pair_f0, pair_f1, seeds, stereo_key=None):
    """One device pass over a list of frame pairs, reduced to what loop-closure validation hands on (:1136-1143): per
    pair the counters, and per supporter ONE 12-word row -- key-point indices (L0, R0, L1, R1) and, as float32 bit
    patterns, the eight coordinates (xL0, yL0, xR0, yR0, xL1, yL1, xR1, yR1) its two Features are built from.
    -> (pose float64[P,3,4], stats int32[P,8], rows int32[M,12], rows per pair int64[P])."""
    pair_f0, pair_f1 = np.asarray(pair_f0, np.int32), np.asarray(pair_f1, np.int32)
    res, xy, _ = track_pairs_arrays(frames, pair_f0, pair_f1, seeds, host_kp=False, stereo_key=stereo_key)
    flat, ptr = ops.compact_supporters(res.mask, res.stats[:, 0], res.tracked_kp)
    cnt = np.diff(ptr)
    g0, g1 = np.repeat(pair_f0, cnt), np.repeat(pair_f1, cnt)
    rows = np.empty((len(flat), 12), np.int32)
    rows[:, :4] = flat
    coor = rows[:, 4:].view(np.float32)
    coor[:, 0:2], coor[:, 2:4] = xy[g0, 0, flat[:, 0]], xy[g0, 1, flat[:, 1]]
    coor[:, 4:6], coor[:, 6:8] = xy[g1, 0, flat[:, 2]], xy[g1, 1, flat[:, 3]]
    return res.pose, res.stats, rows, cnt


LOOP_SHARD_PLAN = "auto"      # slam_amd.dist.track_pairs_sharded's plan for the loop-closure doors: "auto", "one" or "two"


def _loop_pair_rows(frames, pair_f0, pair_f1, seeds):
    """_pair_supporter_rows of the whole pair list: in one process, or -- when the caller has initialised
    torch.distributed with more than one rank -- block-partitioned over the ranks (slam_amd.dist.track_pairs_sharded:
    each rank reads and tracks only the frames its block touches; a frame's stereo match is computed by every rank that
    needs it, or -- when the candidates are spread so widely that this repeats much of the work -- once in the whole
    job, the keys all-gathered; seeds are rank 0's; the results are all-gathered, so every rank returns the whole list's
    arrays, equal to the one-process call's).
    -> (stats int32[P,8], rows int32[M,12], row pointer int64[P+1])."""
    import torch.distributed as tdist
    frames = np.asarray(frames)
    P = len(pair_f0)
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        from . import dist as sdist
        if seeds is None and tdist.get_rank() == 0:
            seeds = np.random.randint(0, 2 ** 32, P, dtype=np.uint64)      # one draw from the global stream per pair
        _, stats, rows, ptr = sdist.track_pairs_sharded(
            pair_f0, pair_f1, seeds, lambda used, l0, l1, sd, sk: _pair_supporter_rows(frames[used], l0, l1, sd, sk), 12,
            n_frames=len(frames), stereo_block=lambda fa, fb: stereo_keys_of(frames[fa:fb]), plan=LOOP_SHARD_PLAN)
    else:
        _, stats, rows, cnt = _pair_supporter_rows(frames, pair_f0, pair_f1, seeds)
        ptr = np.zeros(P + 1, np.int64)
        np.cumsum(cnt, out=ptr[1:])
    if (stats[:, 5] & _lib.SLAM_PAIR_EMPTY).any():
        raise ValueError("low >= high")          # np.random.randint(0, 0, 4) on an empty tracked set (:502)
    return stats, rows, ptr


def _little_tracks_of_rows(rows, first_frame_id, second_frame_id):
    """create_little_tracks (:1149-1167) of the supporters' rows of one pair (see _pair_supporter_rows)."""
    kp, c = rows[:, :4].tolist(), rows[:, 4:].view(np.float32).tolist()
    f0, f1 = np.empty(len(kp), dtype=object), np.empty(len(kp), dtype=object)
    for i, (k, x) in enumerate(zip(kp, c)):
        f0[i] = Feature(k[0], k[1], x[0], x[2], x[1], x[3])
        f1[i] = Feature(k[2], k[3], x[4], x[6], x[5], x[7])
    return create_little_tracks(f0, f1, first_frame_id, second_frame_id)


def find_loop_cand_by_consensus(mahalanobis_dist_cand_movie_ind, mahalanobis_dist_cand_pg_ind, cur_frame_movie_ind,
                                threshold, seeds=None):
    """:1118-1146.  All candidates of the current frame go through the device in one pass (each frame's detections
    and stereo match are used once instead of being recomputed per candidate, :1126)."""
    cands = [int(c) for c in mahalanobis_dist_cand_movie_ind]
    if not cands:
        return []
    frames = np.array([int(cur_frame_movie_ind)] + cands)
    pair_f0 = np.arange(1, len(frames), dtype=np.int32)            # candidate = "frame 0" of the pair (:1128-1134)
    pair_f1 = np.zeros(len(cands), np.int32)                       # current frame = "frame 1"
    stats, rows, ptr = _loop_pair_rows(frames, pair_f0, pair_f1, seeds)
    passed = []
    for c in range(len(cands)):
        if 100 * int(stats[c, 1]) / int(stats[c, 0]) > threshold:  # strict (:1141)
            passed.append([mahalanobis_dist_cand_pg_ind[c],
                           _little_tracks_of_rows(rows[ptr[c]:ptr[c + 1]], cands[c], cur_frame_movie_ind)])
    return passed


def find_loop_cands_by_consensus_batch(cand_movie_ind_lists, cand_pg_ind_lists, cur_frame_movie_inds, threshold, seeds=None):
    """find_loop_cand_by_consensus (:1118-1146) for MANY current frames in one device pass: what the key-frame loop of
    PoseGraph.loop_closure (PoseGraph.py:463-472 -> :386-410) asks for one frame at a time.  Every frame that occurs,
    as current frame or as candidate, is detected and stereo-matched once; all (candidate, current) pairs go through
    one slam_track_pairs call.  Returns one `[[pg_index, tracks], ...]` list per current frame, each equal to what
    find_loop_cand_by_consensus returns for that frame with the same per-pair seeds (`seeds`: one per pair, in the
    order current frame by current frame, candidate by candidate).

    Under torch.distributed with more than one rank (one process per GPU, initialised by the caller) the flat pair list
    is block-partitioned over the ranks (_loop_pair_rows): the candidates of the key-frame loop are validated on all
    GPUs of the node, and every rank returns the whole result."""
    curs = [int(c) for c in cur_frame_movie_inds]
    lists = [[int(c) for c in cl] for cl in cand_movie_ind_lists]
    if len(lists) != len(curs) or len(cand_pg_ind_lists) != len(curs):
        raise ValueError("one candidate list and one pose-graph index list per current frame")
    frames = sorted(set(curs).union(*map(set, lists))) if curs else []
    pos = {f: i for i, f in enumerate(frames)}
    pair_f0 = np.array([pos[c] for cl in lists for c in cl], np.int32)                   # candidate = "frame 0" (:1128-1134)
    pair_f1 = np.array([pos[cur] for cur, cl in zip(curs, lists) for _ in cl], np.int32)  # current frame = "frame 1"
    out = [[] for _ in curs]
    if len(pair_f0) == 0:
        return out
    stats, rows, ptr = _loop_pair_rows(np.array(frames), pair_f0, pair_f1, seeds)
    p = 0
    for k, (cur, cl) in enumerate(zip(curs, lists)):
        for c, cand in enumerate(cl):
            if 100 * int(stats[p, 1]) / int(stats[p, 0]) > threshold:                    # strict (:1141)
                out[k].append([cand_pg_ind_lists[k][c], _little_tracks_of_rows(rows[ptr[p]:ptr[p + 1]], cand, cur)])
            p += 1
    return out

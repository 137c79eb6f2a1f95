"""The reference-shaped entry points (slam_amd.utills) against the reference's golden outputs and the oracle.
These read like the reference's own call sites: same names, positional arguments and return types."""
import numpy as np
import pytest

import slam_oracle as orc
from slam_amd import types
from slam_amd import utills as U

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5                                                    # north_star: poses / points within 1e-5 relative


@pytest.fixture()
def movie(golden):
    src = U.ArrayDetections(golden["movie_desc"], golden["movie_xy"], golden["movie_count"])
    U.set_detection_source(src)
    yield src
    U.set_detection_source(None)


def _triples(ms):
    return np.array([[m.queryIdx, m.trainIdx, m.distance] for m in ms], np.int64).reshape(-1, 3)


def test_match_and_rect_rejection_equal_reference(golden, movie):
    l_kpts, l_dsc, r_kpts, matches, rec = U.read_and_rec_match(2)
    assert isinstance(matches, np.ndarray) and matches.dtype == object
    assert np.array_equal(_triples(matches), golden["pair_m0"])
    assert rec == list(golden["pair_rec0"])
    l1_kpts, l1_dsc, *_ = U.read_and_rec_match(3)
    assert np.array_equal(_triples(U.match(l_dsc, l1_dsc)), golden["pair_tm"])
    # sorted variant and the kNN + ratio path
    srt = U.bf_matching(U.MATCHES_NORM, l_dsc, l1_dsc)
    assert [m.distance for m in srt] == sorted(m.distance for m in srt)
    q, t, d = orc.knn_ratio_match(l_dsc, l1_dsc)
    got = U.match(l_dsc, l1_dsc, U.REGULAR_KNN)
    assert np.array_equal(_triples(got), np.stack([q, t, d], 1))


def test_triangulate_equals_reference(golden):
    X = U.triangulate(U.K @ U.M1, U.K @ U.M2, golden["tri_xl"], golden["tri_xr"])
    err = np.linalg.norm(X - golden["tri_X"], axis=1) / np.linalg.norm(golden["tri_X"], axis=1)
    assert err.max() <= REL_TOL


def test_consensus_match_equals_reference(golden):
    g = golden
    m = U.consensus_match(U.K, g["cons_T"], U.M2, g["cons_X"], g["cons_l1"], g["cons_r1"], 2)
    assert m.dtype == bool and np.array_equal(m, g["cons_mask"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_online_est_pnp_ransac_equals_reference(golden, tag):
    g = golden
    X, l, r = g[f"ransac_{tag}_X"], g[f"ransac_{tag}_l1"], g[f"ransac_{tag}_r1"]
    np.random.seed(int(g[f"ransac_{tag}_seed"]))
    T, mask = U.online_est_pnp_ransac(4, X, U.M1, l, U.M2, r, l, U.M2, r, U.K, acc=2)
    assert np.array_equal(mask, g[f"ransac_{tag}_mask"])                      # supporters bit-exact
    want = g[f"ransac_{tag}_T"]
    assert np.linalg.norm(T - want) / np.linalg.norm(want) <= REL_TOL
    assert np.random.randint(0, 1 << 30) == int(g[f"ransac_{tag}_next_rand"])    # global stream left where the reference leaves it


def test_online_est_pnp_ransac_empty_raises_value_error():
    with pytest.raises(ValueError):
        U.online_est_pnp_ransac(4, np.zeros((0, 3)), U.M1, np.zeros((0, 2)), U.M2, np.zeros((0, 2)), np.zeros((0, 2)), U.M2,
                                np.zeros((0, 2)), U.K, acc=2)


def test_find_features_in_consecutive_frames_equals_reference(golden, movie):
    g = golden
    a, b = U.read_and_rec_match(2), U.read_and_rec_match(3)
    np.random.seed(4242)
    trans, f0, f1, perc, mask = U.find_features_in_consecutive_frames(*a, *b)
    assert np.array_equal(mask, g["pair_mask"]) and perc == float(g["pair_perc"])
    assert np.array_equal([[f.get_left_kpt(), f.get_right_kpt()] for f in f0], g["pair_f0_kp"])
    assert np.array_equal([[f.get_left_kpt(), f.get_right_kpt()] for f in f1], g["pair_f1_kp"])
    assert np.array_equal([[*f.get_left_coor(), *f.get_right_coor()] for f in f0], g["pair_f0_xy"])
    assert np.linalg.norm(trans - g["pair_T"]) / np.linalg.norm(g["pair_T"]) <= REL_TOL
    t2, m2 = U.compute_trans_between_cur_to_next(*a, *b)
    assert m2.shape == mask.shape and t2.shape == (3, 4)


def test_whole_movie_and_database(golden, movie):
    g = golden
    F = len(g["movie_count"])
    seeds = 5000 + np.arange(F - 1)
    feats, percs, glob, rel = U.find_features_in_consecutive_frames_whole_movie(U.M1, seeds=seeds)
    assert len(feats) == F - 1 and len(rel) == F and len(glob) == F and np.array_equal(rel[0], U.M1)

    def frame(i):
        n0, n1 = g["movie_count"][i]
        return dict(dsc_l=g["movie_desc"][i, 0, :n0, :61], dsc_r=g["movie_desc"][i, 1, :n1, :61],
                    xy_l=g["movie_xy"][i, 0, :n0], xy_r=g["movie_xy"][i, 1, :n1])

    want_rel = [orc.M1]
    for p in range(F - 1):
        np.random.seed(int(seeds[p]))
        w = orc.track_pair(frame(p), frame(p + 1))
        sup = w["kp"][w["mask"]]
        assert np.array_equal([[f.get_left_kpt(), f.get_right_kpt()] for f in feats[p][0]], sup[:, :2])
        assert np.array_equal([[f.get_left_kpt(), f.get_right_kpt()] for f in feats[p][1]], sup[:, 2:])
        assert percs[p] == w["inliers_perc"]
        assert np.linalg.norm(rel[p + 1] - w["T"]) / np.linalg.norm(w["T"]) <= REL_TOL
        want_rel.append(w["T"])
    # trajectory: camera centres of the composed poses within 0.1 % of the oracle's (north_star ATE bound)
    c_got = U.left_cameras_trajectory(glob)
    c_want = orc.camera_centres(orc.rel_to_global(want_rel))
    ate = np.sqrt(np.mean(np.sum((c_got - c_want) ** 2, axis=1)))
    assert ate <= 1e-3 * np.linalg.norm(c_want[-1])
    # the feature lists arrive as index arrays (LazyPairFeatures): entry by entry they equal the eagerly built lists ...
    assert isinstance(feats, types.LazyPairFeatures)
    eager = [[U._features_of(s[:, :2], movie.xy[p]), U._features_of(s[:, 2:], movie.xy[p + 1])]
             for p, s in enumerate(feats.kp_pairs)]
    assert [[[f.__getstate__() for f in half] for half in pair] for pair in feats] == \
           [[[f.__getstate__() for f in half] for half in pair] for pair in eager]
    # ... DataBase(...) of the lazy lists (Data.py:43) is the array-backed database, of the eager lists the object one
    assert isinstance(types.DataBase(feats, percs, glob, rel), types.ArrayDataBase)
    db = types.DataBase(eager, percs, glob, rel)
    assert type(db) is types.DataBase and db.get_num_frames() == F and db.get_num_tracks() > 100
    assert max(t.get_track_len() for t in db.get_tracks()) >= 3
    # row N1: the array-backed database built straight from the device index arrays is the same database
    adb = U.whole_movie_database(U.M1, seeds=seeds)
    assert isinstance(adb, types.ArrayDataBase) and adb.get_num_tracks() == db.get_num_tracks()
    assert np.array_equal(adb.get_table().track_len(), [t.get_track_len() for t in db.get_tracks()])
    for fo, fa in zip(db.get_frames(), adb.get_frames()):
        assert np.array_equal(fo.get_tracks_ids(), fa.get_tracks_ids())
        assert np.array_equal(fo.get_ex_cam_mat(), fa.get_ex_cam_mat())
    for tid in (0, db.get_num_tracks() // 2, db.get_num_tracks() - 1):
        to, ta = db.get_tracks()[tid], adb.get_tracks()[tid]
        assert np.array_equal(to.get_frames_ids(), ta.get_frames_ids())
        assert to.get_left_locations_in_all_frames() == ta.get_left_locations_in_all_frames()
        assert to.get_right_locations_in_all_frames() == ta.get_right_locations_in_all_frames()
    assert adb.get_inliers_percentage_per_frame() == percs
    assert np.array_equal(adb.initial_estimate_poses(), db.initial_estimate_poses())
    pa = adb.get_prev_and_next_frame_features()
    assert [f.__getstate__() for f in pa[2][1]] == [f.__getstate__() for f in feats[2][1]]


def test_find_loop_cand_by_consensus(golden, movie):
    out = U.find_loop_cand_by_consensus([3, 4], [30, 40], 5, threshold=50, seeds=np.array([1, 2]))
    assert [o[0] for o in out] == [30, 40]
    tracks = out[1][1]
    assert tracks[0].get_track_len() == 2 and list(tracks[0].get_frames_ids()) == [4, 5]
    none = U.find_loop_cand_by_consensus([3, 4], [30, 40], 5, threshold=100, seeds=np.array([1, 2]))
    assert none == []                                              # strict '>' (:1141)


def test_loop_candidates_through_the_detection_cache(golden, movie):
    """Row N2: consecutive current frames share candidates; a cached source detects each frame once and changes nothing."""
    from slam_amd import ingest

    class Counting:
        def __init__(self, src): self.src, self.calls = src, []
        def __len__(self): return len(self.src)
        def get(self, i):
            self.calls.append(int(i))
            return self.src.get(i)

    def summary(out):
        return [(pg, [(list(t.get_frames_ids()), t.get_left_locations_in_all_frames()) for t in trs]) for pg, trs in out]

    want5 = summary(U.find_loop_cand_by_consensus([1, 2], [10, 20], 5, threshold=50, seeds=np.array([1, 2])))
    want6 = summary(U.find_loop_cand_by_consensus([1, 2], [10, 20], 6, threshold=50, seeds=np.array([3, 4])))
    counting = Counting(movie)
    cache = ingest.CachedDetections(counting, capacity=8)
    U.set_detection_source(cache)
    got5 = summary(U.find_loop_cand_by_consensus([1, 2], [10, 20], 5, threshold=50, seeds=np.array([1, 2])))
    got6 = summary(U.find_loop_cand_by_consensus([1, 2], [10, 20], 6, threshold=50, seeds=np.array([3, 4])))
    assert got5 == want5 and got6 == want6 and len(want5) + len(want6) > 0
    assert sorted(counting.calls) == [1, 2, 5, 6]           # candidates 1 and 2 were detected once, not twice
    assert cache.hits == 2 and cache.misses == 4


def test_supporters_compacted_on_the_device_equal_the_host_compaction(movie):
    """ops.compact_supporters (device gather + one pinned copy, what the whole-movie driver uses) against
    tracks.supporters_flat on a host copy of the same shard: same rows, same order, same row pointer; and a shard
    without a single supporter."""
    import torch
    from slam_amd import ops, pipeline, tracks
    n = len(U.DETECTIONS)
    f0, f1 = pipeline.consecutive_pairs(n)
    res, xy, kp = U.track_pairs_arrays(np.arange(n), f0, f1, seeds=np.arange(n - 1) + 5)
    want_flat, want_ptr = tracks.supporters_flat(res.stats, res.mask, kp)
    words = ops.bool_to_mask_words(res.mask, res.mask.shape[1])
    flat, ptr = ops.compact_supporters(words, res.stats[:, 0], res.tracked_kp)
    assert flat.dtype == np.int32 and np.array_equal(flat, want_flat) and np.array_equal(ptr, want_ptr) and len(flat) > 100
    flat0, ptr0 = ops.compact_supporters(np.zeros_like(words), res.stats[:, 0], res.tracked_kp)
    assert flat0.shape == (0, 4) and not ptr0.any()
    # bits beyond the tracked count are ignored, as on the host
    flat1, ptr1 = ops.compact_supporters(np.full_like(words, -1), res.stats[:, 0], res.tracked_kp)
    assert np.array_equal(np.diff(ptr1), res.stats[:, 0])

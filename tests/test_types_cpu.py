"""CPU: the boundary types (slam_amd.types) against what the reference's own classes produced for the same
features (golden db_* arrays were recorded from the reference's DataBase, tests/golden/make_golden.py)."""
import pickle

import numpy as np
import pytest

from slam_amd import types


def _features(kp, xy):
    out = np.empty(len(kp), dtype=object)
    for i in range(len(kp)):
        out[i] = types.Feature(int(kp[i, 0]), int(kp[i, 1]), xy[i, 0], xy[i, 2], xy[i, 1], xy[i, 3])
    return out


def _db_from_golden(g):
    n_pairs = len(g["movie_perc"])
    feats = [[_features(g[f"movie_f0_kp_{i}"], g[f"movie_f0_xy_{i}"]), _features(g[f"movie_f1_kp_{i}"], g[f"movie_f1_xy_{i}"])]
             for i in range(n_pairs)]
    return types.DataBase(feats, list(g["movie_perc"]), list(g["movie_glob"]), list(g["movie_rel"]))


def test_database_track_chaining_matches_reference(golden):
    g = golden
    db = _db_from_golden(g)
    assert db.get_num_tracks() == int(g["db_num_tracks"])
    assert db.get_num_frames() == len(g["movie_glob"])
    assert np.array_equal([t.get_track_len() for t in db.get_tracks()], g["db_track_len"])
    assert np.array_equal([t.get_first_frame_ind() for t in db.get_tracks()], g["db_track_first"])
    assert np.array_equal([f.get_tracks_num() for f in db.get_frames()], g["db_frame_ntracks"])
    for i, f in enumerate(db.get_frames()):
        assert np.array_equal(f.get_tracks_ids(), g[f"db_frame_tracks_{i}"])
    assert np.array_equal(db.initial_estimate_poses(), g["db_poses"])
    assert np.array_equal(db.get_relative_cam_trans(), g["movie_rel"])


def test_database_pickles(golden, tmp_path):
    db = _db_from_golden(golden)
    path = tmp_path / "db.pickle"
    types.save(str(path), db)
    db2 = types.load(str(path))
    assert db2.get_num_tracks() == db.get_num_tracks()
    t = db2.get_tracks()[3]
    f = t.get_frames_ids()[0]
    assert t.get_feature_at_frame(f).get_left_coor() == db.get_tracks()[3].get_feature_at_frame(f).get_left_coor()
    assert db2.tracks_median_len(0.5) == db.tracks_median_len(0.5)


def test_match_objects_form_1d_object_array():
    ms = [types.DMatch(0, 1, 3), types.DMatch(2, 5, 7)]
    arr = np.array(ms)
    assert arr.dtype == object and arr.shape == (2,)          # utills.py:110 relies on this
    assert arr[[1]][0].trainIdx == 5
    k = types.KeyPoint(1.1, 2.2)
    assert k.pt == (float(np.float32(1.1)), float(np.float32(2.2)))
    assert pickle.loads(pickle.dumps(ms[0])).distance == 3.0


def test_track_and_frame_accessors():
    t = types.Track(7)
    f1, f0 = types.Feature(1, 2, 10., 8., 5., 5.5), types.Feature(3, 4, 11., 9., 6., 6.5)
    t.add_feature(5, f1)
    t.add_feature(4, f0)
    assert t.get_id() == 7 and t.get_track_len() == 2
    assert t.get_first_frame_ind() == 4 and t.get_last_frame_ind() == 5
    assert list(t.get_frames_ids()) == [5, 4]                 # insertion order, like the reference's dict keys
    assert t.get_left_locations_in_all_frames() == [(11., 6.), (10., 5.)]
    assert t.get_right_locations_in_specific_frames([5]) == [(8., 5.5)]
    fr = types.Frame(2, np.hstack([np.eye(3), [[1.], [2.], [3.]]]))
    assert np.array_equal(fr.get_pose(), [-1., -2., -3.]) and fr.get_tracks_num() == 0


def test_small_reference_helpers(tmp_path):
    """The reference module's small host helpers that its callers import from utills (no device work)."""
    import random
    from scipy.spatial.transform import Rotation
    from slam_amd import utills as U
    from slam_amd.types import DMatch, KeyPoint, Track, Feature
    k1 = [KeyPoint(1.5, 2.5), KeyPoint(10, 20)]
    k2 = [KeyPoint(3, 4), KeyPoint(30, 40), KeyPoint(5, 6)]
    ms = [DMatch(1, 2, 7), DMatch(0, 1, 3)]
    assert U.get_matches_coor_and_kpts(ms, k1, k2).tolist() == [[1, 10, 20, 5, 6], [0, 1.5, 2.5, 30, 40]]
    a, b, c, d = U.get_inliers_and_outliers_coor_for_rec(ms[:1], ms[1:], k1, k2)
    assert a.tolist() == [[10, 20]] and b.tolist() == [[5, 6]] and c.tolist() == [[1.5, 2.5]] and d.tolist() == [[30, 40]]
    r = np.array([[0.3], [-0.2], [0.5]])
    T = U.rodriguez_to_mat(r, np.array([[1.0], [2.0], [3.0]]))
    assert np.allclose(T[:, :3], Rotation.from_rotvec(r.ravel()).as_matrix(), atol=1e-15) and T[:, 3].tolist() == [1, 2, 3]
    assert np.array_equal(U.rodriguez_to_mat(np.zeros(3), np.zeros(3))[:, :3], np.eye(3))
    M = np.hstack([np.eye(3), [[1.0], [2.0], [3.0]]])
    assert all(np.array_equal(p, [-1, -2, -3]) for p in U.relative_camera_pos_4(M, M, M, M))
    P = np.array([[3.0, 4.0, 0.0]])
    assert U.euclidean_dist(P).tolist() == [np.sqrt(5.0)] and U.euclidean_dist(P[:, :2], dim="2d").tolist() == [5.0]
    poses = tmp_path / "00.txt"
    poses.write_text("\n".join(" ".join(str(float(i + j)) for j in range(12)) for i in range(5)) + "\n")
    gt = U.get_ground_truth_transformations(str(poses), 3)
    assert len(gt) == 3 and gt[2].shape == (3, 4) and gt[2][0, 0] == 2.0 and gt[1][2, 3] == 12.0
    assert U.left_cameras_trajectory(U.get_ground_truth_transformations(str(poses), 5)).shape == (5, 3)
    with pytest.raises(FileNotFoundError):
        U.get_ground_truth_transformations()
    l0 = np.array([[100.0, 5], [50.0, 5], [10.0, 5]]); r0 = np.array([[90.0, 5], [48.0, 5], [12.0, 5]])
    assert U.far_or_neg_pts_rej(l0, r0).tolist() == [True, False, False]
    tr = []
    for i, n in enumerate((2, 3, 2)):
        t = Track(i)
        for f in range(n):
            t.add_feature(f, Feature(0, 0, 0.0, 0.0, 0.0, 0.0))
        tr.append(t)
    random.seed(1)
    assert U.get_rand_track(3, tr).get_id() == 1
    assert abs(U.mahalanobis_dist(np.array([3.0, 4.0]), np.diag([9.0, 16.0])) - np.sqrt(2.0)) < 1e-15


def test_lazy_object_arrays_are_real_object_arrays_made_on_demand():
    """KeyPointArray / DMatchArray / FeatureArray (what read_and_rec_match and find_features_in_consecutive_frames hand
    out): numpy object arrays to every caller -- isinstance, dtype, len, int / slice / mask / index-array access,
    iteration, sorted(), tolist(), copy(), pickle, numpy functions -- with the elements built when first read."""
    import pickle
    q, t, d = np.array([0, 2, 5, 9]), np.array([1, 3, 4, 7]), np.array([10, 20, 30, 5])
    m = types.DMatchArray(q, t, d)
    assert isinstance(m, np.ndarray) and m.dtype == object and m.shape == (4,) and len(m) == 4
    assert int(m._filled.sum()) == 0                                     # nothing built yet
    assert (m[1].queryIdx, m[1].trainIdx, m[1].distance) == (2, 3, 20.0) and isinstance(m[1].distance, float)
    assert m[-1].queryIdx == 9 and int(m._filled.sum()) == 2
    assert m[1] is m[1]                                                  # built once
    with pytest.raises(IndexError):
        m[4]
    sub = m[[3, 0]]                                                      # the fancy index of utills.py:960
    assert isinstance(sub, types.DMatchArray) and [x.queryIdx for x in sub] == [9, 0]
    assert [x.trainIdx for x in m[np.array([True, False, True, False])]] == [1, 4]
    assert [x.queryIdx for x in m[1:3]] == [2, 5] and len(m[:0]) == 0
    assert [x.distance for x in sorted(m, key=lambda x: x.distance)] == [5.0, 10.0, 20.0, 30.0]     # bf_matching(sort=True)
    assert np.array(m.materialise()).shape == (4,) and np.array(list(m), dtype=object).shape == (4,)
    assert [x.queryIdx for x in m.tolist()] == [0, 2, 5, 9] and m.copy()[2].queryIdx == 5
    assert [x.queryIdx for x in pickle.loads(pickle.dumps(m))] == [0, 2, 5, 9]
    assert np.concatenate([m, m]).shape == (8,) and np.concatenate([m, m])[5].queryIdx == 2
    raw = np.array(types.DMatchArray(q, t, d))                               # a raw copy made behind the array's back ...
    with pytest.raises(RuntimeError):
        raw[0].queryIdx                                                  # ... fails loudly instead of handing out junk
    k = types.KeyPointArray(np.array([[1.5, 2.25], [3.1, 4.2]], np.float32))
    assert k[0].pt == (1.5, 2.25) and k[1].pt == (float(np.float32(3.1)), float(np.float32(4.2)))
    assert isinstance(k[1].pt[0], float)
    f = types.FeatureArray([[1, 2], [3, 4]], np.array([[10, 11, 12, 13], [20, 21, 22, 23]], np.float32))
    assert f[1].get_left_kpt() == 3 and f[1].get_right_kpt() == 4
    assert f[1].get_left_coor() == (20.0, 21.0) and f[1].get_right_coor() == (22.0, 23.0)
    sel = f[np.array([False, True])]
    assert isinstance(sel, types.FeatureArray) and sel[0].__getstate__() == f[1].__getstate__()
    f[0] = "x"                                                           # assignment works like on any object array
    assert f[0] == "x" and f[1].get_left_kpt() == 3

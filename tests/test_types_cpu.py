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

"""Array-backed track chaining / ArrayDataBase against the object DataBase, which
tests/test_types_cpu.py holds to vectors recorded from the reference's DataBase.  No GPU."""
import os
import random

import numpy as np
import pytest

from slam_amd import tracks
from slam_amd.types import ArrayDataBase, DataBase, Feature

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))


def _features(kp, xy_f, cols):
    xl, xr = xy_f[0], xy_f[1]
    out = np.empty(len(kp), dtype=object)
    for j in range(len(kp)):
        l, r = int(kp[j, cols[0]]), int(kp[j, cols[1]])
        out[j] = Feature(l, r, float(xl[l, 0]), float(xr[r, 0]), float(xl[l, 1]), float(xr[r, 1]))
    return out


def _both(kp_pairs, xy_frames):
    n = len(kp_pairs) + 1
    glob = [np.hstack([np.eye(3), np.array([[0.0], [0.0], [-1.0 * f]])]) for f in range(n)]
    feats = [[_features(kp, xy_frames[i], (0, 1)), _features(kp, xy_frames[i + 1], (2, 3))] for i, kp in enumerate(kp_pairs)]
    percs = [50.0] * len(kp_pairs)
    obj = DataBase(feats, percs, glob, glob)
    arr = ArrayDataBase(tracks.chain_tracks(kp_pairs, xy_frames), kp_pairs, xy_frames, percs, glob, glob)
    return obj, arr


def _assert_same(obj, arr):
    assert arr.get_num_tracks() == obj.get_num_tracks() and arr.get_num_frames() == obj.get_num_frames()
    t = arr.get_table()
    assert np.array_equal(t.track_len(), [x.get_track_len() for x in obj.get_tracks()])
    if obj.get_num_tracks():
        assert np.array_equal(t.track_first_frame(), [x.get_first_frame_ind() for x in obj.get_tracks()])
    for fo, fa in zip(obj.get_frames(), arr.get_frames()):
        assert np.array_equal(fo.get_tracks_ids(), fa.get_tracks_ids()) and fo.get_tracks_num() == fa.get_tracks_num()
        assert fa.get_tracks_ids().dtype == fo.get_tracks_ids().dtype
    for to, ta in zip(obj.get_tracks(), arr.get_tracks()):
        assert to.get_id() == ta.get_id() and np.array_equal(to.get_frames_ids(), ta.get_frames_ids())
        assert to.get_left_locations_in_all_frames() == ta.get_left_locations_in_all_frames()
        assert to.get_right_locations_in_all_frames() == ta.get_right_locations_in_all_frames()
        assert to.get_last_frame_ind() == ta.get_last_frame_ind()
        for f in to.get_frames_ids():
            a, b = to.get_feature_at_frame(f), ta.get_feature_at_frame(f)
            assert (a.get_left_kpt(), a.get_right_kpt()) == (b.get_left_kpt(), b.get_right_kpt())
            assert type(b.get_left_kpt()) is int and type(b.get_left_coor()[0]) is float
    assert np.array_equal(obj.initial_estimate_poses(), arr.initial_estimate_poses())


def _golden_pairs():
    n = len(G["movie_rel"]) - 1
    kp = [np.hstack([G[f"movie_f0_kp_{i}"], G[f"movie_f1_kp_{i}"]]) for i in range(n)]
    return kp, [G["movie_xy"][f] for f in range(n + 1)]


def test_golden_movie_matches_the_reference_database():
    kp, xy = _golden_pairs()
    t = tracks.chain_tracks(kp, xy)
    assert t.n_tracks == int(G["db_num_tracks"])
    assert np.array_equal(t.track_len(), G["db_track_len"]) and np.array_equal(t.track_first_frame(), G["db_track_first"])
    assert np.array_equal(np.diff(t.frame_ptr), G["db_frame_ntracks"])
    for f in range(t.n_frames):
        assert np.array_equal(t.frame_tracks[t.frame_ptr[f]:t.frame_ptr[f + 1]], G[f"db_frame_tracks_{f}"])
    # coordinates carried by the table are the reference's Feature coordinates
    for i in range(len(kp)):
        rows = np.nonzero(t.obs_frame == i + 1)[0]
        got = {(int(a), int(b)): tuple(c) for (a, b), c in zip(t.obs_kp[rows], t.obs_xy[rows])}
        for (a, b), (xl, yl, xr, yr) in zip(G[f"movie_f1_kp_{i}"], G[f"movie_f1_xy_{i}"]):
            assert got[(int(a), int(b))] == (xl, xr, yl, yr)
    obj, arr = _both(kp, xy)
    _assert_same(obj, arr)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_chains_equal_the_object_database(seed):
    rng = np.random.default_rng(seed)
    n_frames, n_kp = 25, 60
    xy = [rng.random((2, n_kp, 2)).astype(np.float32) * 1000 for _ in range(n_frames)]
    kp_pairs = []
    for i in range(n_frames - 1):
        n = int(rng.integers(0, 40)) if i != 7 else 0                 # one pair without supporters breaks every track
        l0 = np.sort(rng.choice(n_kp, n, replace=False))
        l1 = rng.choice(n_kp, n, replace=False)                       # injective, arbitrary order
        kp_pairs.append(np.stack([l0, rng.integers(0, n_kp, n), l1, rng.integers(0, n_kp, n)], 1).astype(np.int32).reshape(-1, 4))
    obj, arr = _both(kp_pairs, xy)
    _assert_same(obj, arr)
    # the reference's accessors on top
    lens = arr.get_table().track_len()
    if len(lens):
        L = int(lens[0])
        assert [t.get_id() for t in arr.get_tracks_with_specific_len(L)] == [t.get_id() for t in obj.get_tracks_with_specific_len(L)]
        assert arr.tracks_median_len(0.5) == obj.tracks_median_len(0.5)
        random.seed(5); a = obj.get_rand_track(L).get_id()
        random.seed(5); b = arr.get_rand_track(L).get_id()
        assert a == b and random.random() == (random.seed(5), obj.get_rand_track(L), random.random())[2]
        f = 3
        assert [t.get_id() for t in arr.get_tracks_at_frame(f)] == [t.get_id() for t in obj.get_tracks_at_frame(f)]
        assert [x.get_id() for x in arr.get_frames_at_track(0)] == [x.get_id() for x in obj.get_frames_at_track(0)]
        with pytest.raises(ValueError):
            arr.get_rand_track(10 ** 6)
    pf_o, pf_a = obj.get_prev_and_next_frame_features(), arr.get_prev_and_next_frame_features()
    assert len(pf_o) == len(pf_a)
    for i in (0, 7, len(pf_o) - 1):
        for side in range(2):
            assert [x.__getstate__() for x in pf_o[i][side]] == [x.__getstate__() for x in pf_a[i][side]]


def test_duplicates_are_refused_not_mischained():
    xy = [np.zeros((2, 8, 2), np.float32)] * 3
    dup_first = [np.array([[1, 0, 2, 0], [1, 0, 3, 0]]), np.array([[2, 0, 4, 0]])]
    dup_second = [np.array([[1, 0, 2, 0], [5, 0, 2, 0]]), np.array([[2, 0, 4, 0]])]
    for pairs in (dup_first, dup_second):
        assert not tracks.pairs_are_chainable(pairs)
        with pytest.raises(ValueError):
            tracks.chain_tracks(pairs, xy)


def test_empty_movie_and_supporters_from_shard():
    t = tracks.chain_tracks([np.empty((0, 4), np.int32)], [np.zeros((2, 4, 2), np.float32)] * 2)
    assert t.n_tracks == 0 and t.n_frames == 2 and len(t.frame_tracks) == 0
    stats = np.array([[3, 2, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 2, 0, 0]], np.int32)
    mask = np.array([[True, False, True, True], [False] * 4])
    kp = np.arange(2 * 4 * 4).reshape(2, 4, 4)
    s = tracks.supporters_from_shard(stats, mask, kp)
    assert np.array_equal(s[0], kp[0, [0, 2]]) and s[1].shape == (0, 4)


def test_array_database_saves_and_loads(tmp_path):
    """DataBase.save / load (DataBase.py:15-34) on the array-backed database, before and after objects were made."""
    from slam_amd import types
    kp, xy = _golden_pairs()
    obj, arr = _both(kp, xy)
    _ = arr.get_tracks()[3]                                  # one materialised track in the cache
    path = str(tmp_path / "db.pkl")
    types.save(path, arr)
    back = types.load(path)
    _assert_same(obj, back)
    assert np.array_equal(back.get_relative_cam_trans(), arr.get_relative_cam_trans())


@pytest.mark.parametrize("seed", [0, 3])
def test_native_chaining_equals_the_numpy_restatement_field_by_field(seed):
    """slam_chain_tracks (csrc/tracks_host.cpp) against the per-pair numpy version of the same rule on a movie-sized
    case: ragged pairs, empty pairs, long tracks; coordinates taken from one [F, 2, n_max, 2] block or a list of frames."""
    rng = np.random.default_rng(seed)
    n_frames, n_kp = 300, 2048
    xy = rng.random((n_frames, 2, n_kp, 2)).astype(np.float32) * 1200
    kp_pairs, prev = [], np.empty(0, np.int64)
    for i in range(n_frames - 1):
        n = int(rng.integers(0, 1500)) if i % 97 != 50 else 0
        # most supporters continue a key-point the previous pair ended on, as in a real movie
        keep = rng.permutation(prev)[:int(0.8 * min(n, len(prev)))]
        fresh = rng.permutation(np.setdiff1d(np.arange(n_kp), keep))[:n - len(keep)]
        l0 = np.sort(np.concatenate([keep, fresh]))
        l1 = rng.choice(n_kp, len(l0), replace=False)
        kp_pairs.append(np.stack([l0, rng.integers(0, n_kp, len(l0)), l1, rng.integers(0, n_kp, len(l0))], 1).astype(np.int32).reshape(-1, 4))
        prev = l1
    want = tracks.chain_tracks_numpy(kp_pairs, xy)
    for got in (tracks.chain_tracks(kp_pairs, xy), tracks.chain_tracks(kp_pairs, [xy[f] for f in range(n_frames)]),
                tracks.chain_tracks(kp_pairs, xy, flat=tracks.flatten_pairs(kp_pairs))):
        assert (got.n_frames, got.n_tracks) == (want.n_frames, want.n_tracks) and want.track_len().max() > 5
        for name in ("track_ptr", "obs_frame", "obs_kp", "obs_xy", "frame_ptr", "frame_tracks"):
            a, b = getattr(got, name), getattr(want, name)
            assert a.dtype == b.dtype and np.array_equal(a, b), name


def test_native_chaining_refuses_bad_indices():
    xy = np.zeros((2, 2, 8, 2), np.float32)
    with pytest.raises(IndexError):
        tracks.chain_tracks([np.array([[1, 0, 9, 0]])], xy)
    with pytest.raises(IndexError):
        tracks.chain_tracks([np.array([[1, 0, 2, -1]])], xy)
    with pytest.raises(ValueError):
        tracks.chain_tracks([np.array([[1, 0, 2, 0]])], xy, flat=(np.array([[1, 0, 2, 0]], np.int32), np.array([0, 2])))


def test_array_database_array_file_round_trip(tmp_path):
    """ArrayDataBase.save_arrays / load_arrays: a file of plain arrays (np.load with allow_pickle=False), equal to the
    database it was written from, with a pair whose pose is None."""
    kp, xy = _golden_pairs()
    n = len(kp) + 1
    glob = [np.hstack([np.eye(3), np.array([[0.0], [0.0], [-1.0 * f]])]) for f in range(n)]
    rel = [np.hstack([np.eye(3), np.array([[0.0], [0.0], [-1.0]])]) for _ in range(n)]
    rel[3] = None
    percs = [50.0 + i for i in range(len(kp))]
    arr = ArrayDataBase(tracks.chain_tracks(kp, xy), kp, xy, percs, glob, rel)
    path = str(tmp_path / "db.npz")
    arr.save_arrays(path)
    with np.load(path, allow_pickle=False) as z:                     # nothing in the file needs pickle
        assert set(z.files) >= {"supporters", "pair_ptr", "xy", "global_trans", "relative_trans", "inliers_percentage"}
    back = ArrayDataBase.load_arrays(path)
    obj, _ = _both(kp, xy)
    _assert_same(obj, back)
    t0, t1 = arr.get_table(), back.get_table()
    for name in ("track_ptr", "obs_frame", "obs_kp", "obs_xy", "frame_ptr", "frame_tracks"):
        assert np.array_equal(getattr(t0, name), getattr(t1, name)), name
    r0, r1 = arr.get_relative_cam_trans(), back.get_relative_cam_trans()
    assert len(r0) == len(r1) and r1[3] is None and all(np.array_equal(a, b) for i, (a, b) in enumerate(zip(r0, r1)) if i != 3)
    assert list(back._inliers_percentage) == percs
    with pytest.raises(ValueError):
        np.savez(str(tmp_path / "other.npz"), format=np.array("something else"))
        ArrayDataBase.load_arrays(str(tmp_path / "other.npz"))


_NOMEM_SCRIPT = r"""
import ctypes, resource, sys
L = ctypes.CDLL(sys.argv[1])
L.slam_last_error_string.restype = ctypes.c_char_p
vm_kb = int([l for l in open('/proc/self/status') if l.startswith('VmSize')][0].split()[1])
limit = vm_kb * 1024 + (256 << 20)
resource.setrlimit(resource.RLIMIT_AS, (limit, limit))
ptr = (ctypes.c_int64 * 1)(0)
track_ptr = (ctypes.c_int64 * 2)()
frame_ptr = (ctypes.c_int64 * 2)()
nt, no = ctypes.c_int64(-1), ctypes.c_int64(-1)
rc = L.slam_chain_tracks(None, ptr, 0, 1 << 28, None, 0, track_ptr, None, None, None, frame_ptr, None,
                         ctypes.byref(nt), ctypes.byref(no))
print(rc, L.slam_last_error_string().decode())
"""


def test_chain_tracks_reports_host_out_of_memory_instead_of_terminating():
    """No C++ exception crosses the C ABI: with the address space capped, the 1 GiB key-point tables of a 2^28-row image
    cannot be allocated and the call returns SLAM_E_HOST with a message (it used to end the process in std::terminate)."""
    import subprocess
    import sys

    from slam_amd import _lib
    p = subprocess.run([sys.executable, "-c", _NOMEM_SCRIPT, _lib.LIB_PATH], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    rc, msg = p.stdout.strip().split(" ", 1)
    assert int(rc) == _lib.SLAM_E_HOST and "out of host memory" in msg

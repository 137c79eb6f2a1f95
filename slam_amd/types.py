"""Plain-Python types that cross the hot-path boundary, with the reference's signatures.

* `DMatch` / `KeyPoint`: duck types of cv2.DMatch / cv2.KeyPoint.  The reference reads only
  `.queryIdx/.trainIdx/.distance` (utils/utills.py:69,147,161-162,433,453-454,969) and `.pt` (:163-164).
  They are classes with __slots__, NOT tuples: `np.array(matches)` (utills.py:110) must stay a 1-D object array.
* `Feature` (DataBaseDirectory/Feature.py:3-39), `Frame` (Frame.py:5-58), `Track` (Trck.py:5-105),
  `DataBase` (DataBase.py:37-328): same constructors and accessors, re-implemented; they hold only Python/numpy
  values, so a DataBase can be saved and loaded with pickle the way the reference does (`save` / `load`,
  DataBase.py:15-34).  The FILES are not interchangeable with the reference's db_*.pickle files: a pickle names the
  defining module and attributes (`DataBaseDirectory.DataBase.DataBase`, `_DataBase__tracks` there;
  `slam_amd.types.DataBase`, `_tracks` here), so each side loads only what it saved.  `load` unpickles: use it on
  files you wrote yourself, never on untrusted ones.
"""
from __future__ import annotations

import bisect
import pickle
import random

import numpy as np

from .synth import K_KITTI


class DMatch:
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx=-1, trainIdx=-1, distance=float("inf"), imgIdx=0):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return f"DMatch(queryIdx={self.queryIdx}, trainIdx={self.trainIdx}, distance={self.distance})"

    def __getstate__(self):
        return (self.queryIdx, self.trainIdx, self.imgIdx, self.distance)

    def __setstate__(self, s):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = s


class KeyPoint:
    __slots__ = ("pt",)

    def __init__(self, x, y):
        self.pt = (float(np.float32(x)), float(np.float32(y)))     # cv2 stores float32, returns Python floats

    def __repr__(self):
        return f"KeyPoint{self.pt}"

    def __getstate__(self):
        return self.pt

    def __setstate__(self, s):
        self.pt = s


def keypoints_from_xy(xy) -> np.ndarray:
    """float[n, 2] -> object array of KeyPoint (what np.array(alg.detectAndCompute(...)[0]) yields, utills.py:54)."""
    a = np.asarray(xy, dtype=np.float32).reshape(-1, 2).astype(np.float64)     # cv2 stores float32, returns Python floats
    out = np.empty(len(a), dtype=object)
    new = KeyPoint.__new__
    for i, pt in enumerate(zip(a[:, 0].tolist(), a[:, 1].tolist())):           # one conversion for all, not two per point
        k = new(KeyPoint)
        k.pt = pt
        out[i] = k
    return out


def keypoints_to_xy(kpts) -> np.ndarray:
    """object array / list of things with `.pt` -> float32[n, 2]."""
    out = np.empty((len(kpts), 2), np.float32)
    for i, k in enumerate(kpts):
        out[i, 0], out[i, 1] = k.pt
    return out


def dmatches_from_arrays(q, t, d) -> np.ndarray:
    out = np.empty(len(q), dtype=object)
    new = DMatch.__new__
    rows = zip(np.asarray(q, np.int64).tolist(), np.asarray(t, np.int64).tolist(), np.asarray(d, np.float64).tolist())
    for i, (qi, ti, di) in enumerate(rows):
        m = new(DMatch)
        m.queryIdx, m.trainIdx, m.imgIdx, m.distance = qi, ti, 0, di
        out[i] = m
    return out


class Feature:
    """A feature of a stereo frame: key-point indices and pixel coordinates in the left and right image."""
    __slots__ = ("_kl", "_kr", "_xl", "_xr", "_yl", "_yr")

    def __init__(self, kpt_left_idx, kpt_right_idx, x_left, x_right, y_left, y_right):
        self._kl, self._kr = kpt_left_idx, kpt_right_idx
        self._xl, self._xr, self._yl, self._yr = x_left, x_right, y_left, y_right

    def get_left_coor(self):
        return self._xl, self._yl

    def get_right_coor(self):
        return self._xr, self._yr

    def get_left_kpt(self):
        return self._kl

    def get_right_kpt(self):
        return self._kr

    def __getstate__(self):
        return (self._kl, self._kr, self._xl, self._xr, self._yl, self._yr)

    def __setstate__(self, s):
        self._kl, self._kr, self._xl, self._xr, self._yl, self._yr = s


class Frame:
    """A frame: id, the ids of the tracks seen in it, global extrinsic [R|t], calibration, camera centre."""

    def __init__(self, id, ex_cam_mat, calib_cam_mat=K_KITTI):
        self._id = id
        self._tracks = []
        self._ex = ex_cam_mat
        self._calib = calib_cam_mat
        self._pose = self.compute_pose(ex_cam_mat)

    def add_track(self, track_id):
        self._tracks.append(track_id)

    def get_id(self):
        return self._id

    def get_tracks_ids(self):
        return np.array(self._tracks).astype(int)

    def get_tracks_num(self):
        return len(self._tracks)

    def get_ex_cam_mat(self):
        return self._ex

    def get_calib_cam_mat(self):
        return self._calib

    def get_pose(self):
        return self._pose

    def compute_pose(self, ex_cam_mat):
        """Camera centre in world coordinates: -R^T t (Frame.py:50-58)."""
        return -1 * ex_cam_mat[:, :3].T @ ex_cam_mat[:, 3]


class Track:
    """A landmark track: {frame id -> Feature} plus the sorted frame ids."""

    def __init__(self, track_id):
        self._id = track_id
        self._by_frame = {}
        self._sorted_ids = []

    def get_id(self):
        return self._id

    def add_feature(self, frame_id, feature):
        self._by_frame[frame_id] = feature
        bisect.insort(self._sorted_ids, frame_id)

    def get_frames_ids(self):
        return np.array(list(self._by_frame.keys())).astype(int)

    def get_feature_at_frame(self, frame_id):
        return self._by_frame[frame_id]

    def get_left_locations_in_all_frames(self):
        return [self._by_frame[f].get_left_coor() for f in self._sorted_ids]

    def get_left_locations_in_specific_frames(self, frames_ind):
        return [self._by_frame[f].get_left_coor() for f in frames_ind]

    def get_right_locations_in_all_frames(self):
        return [self._by_frame[f].get_right_coor() for f in self._sorted_ids]

    def get_right_locations_in_specific_frames(self, frames_ind):
        return [self._by_frame[f].get_right_coor() for f in frames_ind]

    def get_frames_features_dict(self):
        return self._by_frame

    def get_track_len(self):
        return len(self._by_frame)

    def get_last_frame_ind(self):
        return self._sorted_ids[-1]

    def get_first_frame_ind(self):
        return self._sorted_ids[0]


def _pose_array(poses):
    """np.array(poses) as the reference does (DataBase.py:45); a list holding None (a pair without a pose) becomes an
    object array instead of the ValueError numpy >= 1.24 raises for ragged input."""
    try:
        return np.array(poses)
    except ValueError:
        out = np.empty(len(poses), dtype=object)
        for i, p in enumerate(poses):
            out[i] = p
        return out


def save(path, db):
    """DataBase.py:15-22."""
    with open(path, "wb") as fh:
        pickle.dump(db, fh)


def load(saved_db_path):
    """DataBase.py:25-34."""
    with open(saved_db_path, "rb") as fh:
        return pickle.load(fh)


class DataBase:
    """Track database built from the per-pair supporter features of the hot path (DataBase.py:37-153).

    Tracks are chained on the LEFT key-point index: a supporter of pair (i, i+1) whose frame-i left key-point
    ended a track in pair (i-1, i) extends that track; otherwise it opens a new one.  Track ids are assigned in
    encounter order, which is the order of the features inside each pair (ascending left key-point of frame i)."""

    def __new__(cls, prev_and_next_frame_features=None, *args, **kwargs):
        # Feature lists that are still index arrays (what find_features_in_consecutive_frames_whole_movie returns,
        # LazyPairFeatures) go to the array-backed database: same accessors, no 4.5 M Feature objects in between.
        if cls is DataBase and isinstance(prev_and_next_frame_features, LazyPairFeatures):
            return object.__new__(ArrayDataBase)
        return object.__new__(cls)

    def __init__(self, prev_and_next_frame_features, inliers_percentage, global_cam_trans, relative_cam_trans):
        self._tracks = []
        self._frames = []
        self._last_kpts = {}
        self._inliers_percentage = inliers_percentage
        self._pair_features = prev_and_next_frame_features
        self._relative = _pose_array(relative_cam_trans)
        self.create_db(prev_and_next_frame_features, global_cam_trans)

    def create_db(self, prev_and_next_frame_features, global_cam_trans):
        self.add_frame(Frame(0, global_cam_trans[0]))
        for i, (cur, nxt) in enumerate(prev_and_next_frame_features):
            self.set_tracks_and_frames(cur, nxt, i, global_cam_trans)
        self._frames = np.array(self._frames)
        self._tracks = np.array(self._tracks)

    def set_tracks_and_frames(self, first_frame_features_obj, second_frame_features_obj, first_frame_id, global_trans):
        second = first_frame_id + 1
        self.add_frame(Frame(second, global_trans[second]))
        self._last_kpts = self.find_tracks_between_consecutive_frame(first_frame_features_obj,
                                                                     second_frame_features_obj, first_frame_id, second)

    def find_tracks_between_consecutive_frame(self, first_frame_features_obj, second_frame_features_obj,
                                              first_frame_id, second_frame_id):
        carried = {}
        f_first, f_second = self._frames[first_frame_id], self._frames[second_frame_id]
        for feat0, feat1 in zip(first_frame_features_obj, second_frame_features_obj):
            tid = self._last_kpts.get(feat0.get_left_kpt())
            if tid is None:
                tid = len(self._tracks)
                track = Track(tid)
                self.add_track(track)
                track.add_feature(first_frame_id, feat0)
                f_first.add_track(tid)
            else:
                track = self._tracks[tid]
            track.add_feature(second_frame_id, feat1)
            f_second.add_track(tid)
            carried[feat1.get_left_kpt()] = tid
        return carried

    def add_frame(self, frame):
        self._frames.append(frame)

    def add_track(self, track):
        self._tracks.append(track)

    def get_track_features_in_frame(self, frame_id, track_id):
        return self._tracks[track_id].get_feature_at_frame(frame_id)

    def get_num_tracks(self):
        return len(self._tracks)

    def get_num_frames(self):
        return len(self._frames)

    def get_tracks(self):
        return self._tracks

    def get_tracks_at_frame(self, frames_id):
        return self._tracks[self._frames[frames_id].get_tracks_ids()]

    def get_frames_at_track(self, track_id):
        return self._frames[self._tracks[track_id].get_frames_ids()]

    def get_frames(self):
        return self._frames

    def get_inliers_percentage_per_frame(self):
        return self._inliers_percentage

    def get_prev_and_next_frame_features(self):
        return self._pair_features

    def get_rand_track(self, track_len):
        while True:
            t = self._tracks[random.randint(0, len(self._tracks) - 1)]
            if t.get_track_len() == track_len:
                return t

    def get_relative_cam_trans(self):
        return self._relative

    def initial_estimate_poses(self):
        return np.array([f.get_pose() for f in self._frames])

    def get_tracks_with_specific_len(self, track_len):
        return self._tracks[[t.get_track_len() == track_len for t in self._tracks]]

    def tracks_median_len(self, percentage):
        lens = sorted(t.get_track_len() for t in self._tracks)
        return lens[int(percentage * len(lens))]


class _LazyObjects:
    """Sequence of objects made on first access; indexable like the numpy object arrays the reference keeps
    (int, slice, integer array, boolean mask/list), so `tracks[frame.get_tracks_ids()]` keeps working."""

    def __init__(self, n, make):
        self._n, self._make, self._cache = n, make, {}

    def __len__(self):
        return self._n

    def _one(self, i):
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        o = self._cache.get(i)
        if o is None:
            o = self._cache[i] = self._make(i)
        return o

    def __iter__(self):
        return (self._one(i) for i in range(self._n))

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self._one(key)
        if isinstance(key, slice):
            idx = range(*key.indices(self._n))
        else:
            key = np.asarray(key)
            idx = np.nonzero(key)[0] if key.dtype == bool else key.reshape(-1)
        out = np.empty(len(idx), dtype=object)
        for j, i in enumerate(idx):
            out[j] = self._one(i)
        return out

    def materialise(self):
        return self[:]


class _Unset:
    """What a lazily materialised object array holds in a slot nobody has asked for yet.  Reading the array itself (an
    index, a slice, a mask, iteration, tolist, copy, pickle) never shows it; a RAW numpy copy of the buffer made behind
    the array's back (np.array(a), np.asarray(a), a.view(np.ndarray)) does, and using such an element fails loudly."""
    __slots__ = ()

    def __getattr__(self, name):
        raise RuntimeError("element of a lazily materialised array reached through a raw numpy copy of its buffer "
                           "(np.array / np.asarray of the array): index or iterate the array itself, or call "
                           ".materialise() on it first")

    def __repr__(self):
        return "<not materialised>"


_UNSET = _Unset()


class LazyObjectArray(np.ndarray):
    """A real 1-D numpy object array (isinstance np.ndarray, dtype object -- what the reference's callers get from
    np.array(list of cv2 objects), utils/utills.py:54,110,975) whose elements are made on first access from index /
    coordinate arrays kept beside it.  The reference's per-pair functions hand thousands of KeyPoint / DMatch / Feature
    objects from one to the next and read a few fields of each; building them eagerly cost 3 ms per frame pair around
    tens of microseconds of device work.  The hot-path functions of slam_amd.utills read the arrays underneath when
    they are handed one of these, and everything else sees ordinary objects.

    Subclasses set `_n` arrays and implement `_make(idx) -> list of objects` and `_subset(idx) -> same type`."""

    def __new__(cls, n):
        a = np.empty(int(n), dtype=object).view(cls)
        np.ndarray.fill(a, _UNSET)
        a._filled = np.zeros(int(n), bool)
        return a

    def __array_finalize__(self, parent):
        # arrays numpy derives behind our back (views, copies) carry no recipe: their unset slots stay loud (_Unset)
        self._filled = None

    # -- materialisation --
    def _fill(self, idx):
        if self._filled is None:
            return
        idx = np.atleast_1d(idx)
        need = idx[~self._filled[idx]]
        if len(need):
            raw = self.view(np.ndarray)
            for i, o in zip(need.tolist(), self._make(need)):
                raw[i] = o
            self._filled[need] = True

    def materialise(self):
        """The same array with every element made: a plain numpy object array sharing this one's memory."""
        if self._filled is not None and not self._filled.all():
            self._fill(np.nonzero(~self._filled)[0])
        return self.view(np.ndarray)

    # -- everything that reads elements goes through here --
    def __getitem__(self, key):
        if self._filled is None:
            return np.ndarray.__getitem__(self, key)
        if isinstance(key, (int, np.integer)):
            i = int(key)
            if i < 0:
                i += len(self)
            if not 0 <= i < len(self):
                raise IndexError(f"index {key} is out of bounds for axis 0 with size {len(self)}")
            if not self._filled[i]:
                self._fill(i)
            return np.ndarray.__getitem__(self.view(np.ndarray), i)
        if isinstance(key, slice):
            idx = np.arange(*key.indices(len(self)))
        else:
            k = np.asarray(key)
            if k.dtype == bool:
                if k.shape != (len(self),):
                    raise IndexError("boolean index did not match the array")
                idx = np.nonzero(k)[0]
            elif k.ndim == 1 and k.dtype.kind in "iu":
                idx = np.where(k < 0, k + len(self), k).astype(np.int64)
                if len(idx) and (idx.min() < 0 or idx.max() >= len(self)):
                    raise IndexError("index out of bounds")
            elif k.ndim == 1 and k.size == 0:
                idx = np.zeros(0, np.int64)
            else:                                   # anything exotic: plain numpy semantics on the materialised array
                return self.materialise()[key]
        return self._subset(idx)

    def __iter__(self):
        return iter(self.materialise())

    def __setitem__(self, key, value):
        self.materialise()[key] = value

    def tolist(self):
        return self.materialise().tolist()

    def copy(self, order="C"):
        return self.materialise().copy(order)

    def __reduce__(self):
        return self.materialise().__reduce__()     # pickles as the plain object array it stands for

    def __array_function__(self, func, types, args, kwargs):
        # numpy functions (concatenate, array_equal, ...) see the materialised arrays
        def plain(x):
            if isinstance(x, LazyObjectArray):
                return x.materialise()
            if isinstance(x, (list, tuple)):
                return type(x)(plain(v) for v in x)
            return x
        return func(*plain(args), **{k: plain(v) for k, v in kwargs.items()})

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(x.materialise() if isinstance(x, LazyObjectArray) else x for x in inputs)
        if "out" in kwargs:
            kwargs["out"] = tuple(x.materialise() if isinstance(x, LazyObjectArray) else x for x in kwargs["out"])
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __repr__(self):
        done = "?" if self._filled is None else int(self._filled.sum())
        return f"<{type(self).__name__} of {len(self)} ({done} materialised)>"

    __str__ = __repr__


class KeyPointArray(LazyObjectArray):
    """KeyPoint objects over float32 coordinates `xy` [n, 2] (np.array(alg.detectAndCompute(...)[0]), utills.py:54)."""

    def __new__(cls, xy):
        xy = np.ascontiguousarray(np.asarray(xy, dtype=np.float32).reshape(-1, 2))
        a = super().__new__(cls, len(xy))
        a.xy = xy
        return a

    def _make(self, idx):
        out, new = [], KeyPoint.__new__
        for pt in self.xy[idx].astype(np.float64).tolist():
            k = new(KeyPoint)
            k.pt = (pt[0], pt[1])
            out.append(k)
        return out

    def _subset(self, idx):
        return KeyPointArray(self.xy[idx])


class DMatchArray(LazyObjectArray):
    """DMatch objects over (queryIdx, trainIdx, distance) arrays (np.array(matcher.match(...)), utills.py:110)."""

    def __new__(cls, q, t, d):
        q = np.asarray(q, dtype=np.int64).reshape(-1)
        a = super().__new__(cls, len(q))
        a.q, a.t, a.d = q, np.asarray(t, dtype=np.int64).reshape(-1), np.asarray(d, dtype=np.float64).reshape(-1)
        return a

    def _make(self, idx):
        return dmatches_from_arrays(self.q[idx], self.t[idx], self.d[idx]).tolist()

    def _subset(self, idx):
        return DMatchArray(self.q[idx], self.t[idx], self.d[idx])


class FeatureArray(LazyObjectArray):
    """Feature objects (utills.py:963-975) over key-point index pairs `kp` int[n, 2] (left, right) and coordinates `coor`
    float32[n, 4] (x_left, y_left, x_right, y_right)."""

    def __new__(cls, kp, coor):
        kp = np.asarray(kp, dtype=np.int64).reshape(-1, 2)
        a = super().__new__(cls, len(kp))
        a.kp, a.coor = kp, np.asarray(coor, dtype=np.float32).reshape(-1, 4)
        return a

    def _make(self, idx):
        return [Feature(k[0], k[1], c[0], c[2], c[1], c[3])
                for k, c in zip(self.kp[idx].tolist(), self.coor[idx].astype(np.float64).tolist())]

    def _subset(self, idx):
        return FeatureArray(self.kp[idx], self.coor[idx])


class LazyPairFeatures(_LazyObjects):
    """`consecutive_frame_features` of the whole-movie driver (utils/utills.py:867: one [frame-i features, frame-(i+1)
    features] entry per pair) kept as the supporters' key-point index arrays; entry i becomes two object arrays of
    Feature when it is read.  kp_pairs[i]: int[n_i, 4] (L_i, R_i, L_{i+1}, R_{i+1}); xy_frames[f]: float32[2, n, 2]."""

    def __init__(self, kp_pairs, xy_frames, flat=None):
        self.kp_pairs, self.xy_frames = kp_pairs, xy_frames
        self.flat = flat            # optional (int32[M,4], int64[P+1]): the same supporters in one piece, for chain_tracks
        super().__init__(len(kp_pairs), self._make)

    def __getstate__(self):
        state = dict(self.__dict__)
        state["flat"] = None        # kp_pairs holds the same rows; a saved database does not need them twice
        return state

    def _make(self, i):
        kp = self.kp_pairs[i]
        out = []
        for f, cols in ((i, (0, 1)), (i + 1, (2, 3))):
            xl, xr = self.xy_frames[f][0], self.xy_frames[f][1]
            a = np.empty(len(kp), dtype=object)
            for j in range(len(kp)):
                l, r = int(kp[j, cols[0]]), int(kp[j, cols[1]])
                a[j] = Feature(l, r, float(xl[l, 0]), float(xr[r, 0]), float(xl[l, 1]), float(xr[r, 1]))
            out.append(a)
        return out


class ArrayFrame(Frame):
    """Frame whose track ids stay an array until somebody appends to them."""

    def __init__(self, id, ex_cam_mat, track_ids, calib_cam_mat=K_KITTI):
        super().__init__(id, ex_cam_mat, calib_cam_mat)
        self._tracks = track_ids

    def add_track(self, track_id):
        if not isinstance(self._tracks, list):
            self._tracks = self._tracks.tolist()
        self._tracks.append(track_id)


class ArrayDataBase(DataBase):
    """DataBase over a tracks.TrackTable: same accessors as DataBase (DataBase.py:155-328), objects made on demand.

    Built straight from the hot path's index arrays (utills.whole_movie_database); equal, accessor by accessor, to
    DataBase built from the materialised Feature lists."""

    def __init__(self, table, *args):
        """ArrayDataBase(table, kp_pairs, xy_frames, inliers_percentage, global_cam_trans, relative_cam_trans), or the
        reference's four constructor arguments with a LazyPairFeatures first (what DataBase(...) forwards here)."""
        if isinstance(table, LazyPairFeatures):
            from . import tracks
            lazy = table
            inliers_percentage, global_cam_trans, relative_cam_trans = args
            kp_pairs, xy_frames = lazy.kp_pairs, lazy.xy_frames
            table = tracks.chain_tracks(kp_pairs, xy_frames, flat=lazy.flat)
        else:
            lazy = None
            kp_pairs, xy_frames, inliers_percentage, global_cam_trans, relative_cam_trans = args
        self._table = table
        self._kp_pairs, self._xy_frames = kp_pairs, xy_frames
        self._inliers_percentage = inliers_percentage
        self._relative = _pose_array(relative_cam_trans)
        self._last_kpts = {}
        fp, ft = table.frame_ptr, table.frame_tracks
        frames = np.empty(table.n_frames, dtype=object)
        for f in range(table.n_frames):
            frames[f] = ArrayFrame(f, global_cam_trans[f], ft[fp[f]:fp[f + 1]])
        self._frames = frames
        self._tracks = _LazyObjects(table.n_tracks, self._make_track)
        self._pair_features = lazy if lazy is not None else LazyPairFeatures(kp_pairs, xy_frames)

    def _feature(self, row):
        t = self._table
        kl, kr = t.obs_kp[row]
        xl, xr, yl, yr = t.obs_xy[row]
        return Feature(int(kl), int(kr), float(xl), float(xr), float(yl), float(yr))

    def _make_track(self, tid):
        t = self._table
        tr = Track(tid)
        for row in range(t.track_ptr[tid], t.track_ptr[tid + 1]):
            tr.add_feature(int(t.obs_frame[row]), self._feature(row))
        return tr

    def get_table(self):
        return self._table

    def get_num_tracks(self):
        return self._table.n_tracks

    def get_tracks_with_specific_len(self, track_len):
        return self._tracks[self._table.track_len() == track_len]

    def tracks_median_len(self, percentage):
        lens = np.sort(self._table.track_len())
        return int(lens[int(percentage * len(lens))])

    def get_rand_track(self, track_len):
        """DataBase.py:293-305: same draws from `random` as the reference's loop, but only the winner is materialised;
        where the reference would spin forever (no track of that length) this raises."""
        lens = self._table.track_len()
        if not (lens == track_len).any():
            raise ValueError(f"no track of length {track_len}")
        while True:
            idx = random.randint(0, len(lens) - 1)
            if lens[idx] == track_len:
                return self._tracks[idx]

    def add_track(self, track):
        raise TypeError("ArrayDataBase is built whole; use DataBase to grow a database track by track")

    def save_arrays(self, path):
        """The database as one .npz of plain arrays (supporters, coordinates, poses, inlier percentages): a file that
        np.load reads with allow_pickle=False, for callers who do not want to unpickle (`load_arrays` rebuilds it)."""
        from . import tracks
        flat, ptr = self._pair_features.flat or tracks.flatten_pairs(self._kp_pairs)
        xy = tracks._xy_block(self._xy_frames, self._table.n_frames)
        glob = np.array([f.get_ex_cam_mat() for f in self._frames], np.float64)
        rel = self._relative
        rel_ok = np.array([r is not None for r in rel], bool)
        rel_arr = np.array([np.asarray(r, np.float64) if r is not None else np.full((3, 4), np.nan) for r in rel])
        np.savez(path, format=np.array("slam_amd.ArrayDataBase/1"), supporters=flat, pair_ptr=ptr, xy=xy, global_trans=glob,
                 relative_trans=rel_arr, relative_ok=rel_ok, inliers_percentage=np.asarray(self._inliers_percentage, np.float64))

    @staticmethod
    def load_arrays(path):
        from . import tracks
        with np.load(path, allow_pickle=False) as z:
            if str(z["format"]) != "slam_amd.ArrayDataBase/1":
                raise ValueError(f"{path}: not an ArrayDataBase array file")
            flat, ptr, xy = z["supporters"], z["pair_ptr"], z["xy"]
            glob, rel_arr, rel_ok, percs = z["global_trans"], z["relative_trans"], z["relative_ok"], z["inliers_percentage"]
        kp_pairs = [flat[ptr[i]:ptr[i + 1]] for i in range(len(ptr) - 1)]
        rel = [rel_arr[i] if rel_ok[i] else None for i in range(len(rel_arr))]
        lazy = LazyPairFeatures(kp_pairs, xy, flat=(flat, ptr))
        return ArrayDataBase(lazy, percs.tolist(), list(glob), rel)

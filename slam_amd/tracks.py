"""Array-backed track chaining: the immediate consumer of the hot path.

The reference builds its track database pair by pair in Python (DataBaseDirectory/DataBase.py:56-153): for every
supporter of pair (i, i+1), in order, the frame-i LEFT key-point index is looked up in the dictionary left by pair
(i-1, i); a hit extends that track, a miss opens track number len(tracks).  With ~1200 supporters x 3449 pairs that is
millions of Feature / dict operations.  Here the same rule runs on index arrays and produces a CSR table of
observations; Feature / Track / Frame objects are materialised only when somebody asks for them
(slam_amd/types.py:ArrayDataBase keeps the reference's accessors).

chain_tracks is the library's slam_chain_tracks (csrc/tracks_host.cpp: two linear passes over the supporters, a dense
key-point -> track array instead of the per-pair dictionary; 3449 pairs / 4.3 M supporters in tens of milliseconds).
chain_tracks_numpy is the first, per-pair numpy version of the same rule (one searchsorted per pair), kept as the
plain restatement the native one is tested against.

The rule is reproduced exactly, including track numbering (encounter order) and the order of the track ids inside
every frame; tests hold both to the object implementation, which is itself held to vectors recorded from the
reference.  Pure host code."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class TrackTable:
    n_frames: int
    n_tracks: int
    track_ptr: np.ndarray      # int64[n_tracks + 1]: observations of track t are rows track_ptr[t]:track_ptr[t+1]
    obs_frame: np.ndarray      # int32[M]   frame id of every observation, ascending inside a track
    obs_kp: np.ndarray         # int32[M,2] (left, right) key-point index
    obs_xy: np.ndarray         # float64[M,4] x_left, x_right, y_left, y_right (Feature's constructor order)
    frame_ptr: np.ndarray      # int64[n_frames + 1]
    frame_tracks: np.ndarray   # int64[...]  track ids per frame in the reference's insertion order

    def track_len(self) -> np.ndarray:
        return np.diff(self.track_ptr)

    def track_first_frame(self) -> np.ndarray:
        return self.obs_frame[self.track_ptr[:-1]]


def pairs_are_chainable(kp_pairs) -> bool:
    """The array rule needs what the hot path guarantees: inside one pair no frame-i left key-point and no
    frame-(i+1) left key-point appears twice (rect_join emits ascending left0; cross-check is injective into left1)."""
    for kp in kp_pairs:
        if len(kp) and (len(np.unique(kp[:, 0])) != len(kp) or len(np.unique(kp[:, 2])) != len(kp)):
            return False
    return True


def _coords(xy_frame, left, right) -> np.ndarray:
    """float64[n,4] (x_left, x_right, y_left, y_right) of key-point index pairs in one frame."""
    fx = np.asarray(xy_frame, np.float64)
    l, r = fx[0][left], fx[1][right]
    return np.stack([l[:, 0], r[:, 0], l[:, 1], r[:, 1]], 1).reshape(-1, 4)


def flatten_pairs(kp_pairs):
    """list of int[n_i, 4] -> (int32[M, 4] contiguous, int64[P + 1] row pointer)."""
    ptr = np.zeros(len(kp_pairs) + 1, np.int64)
    if len(kp_pairs):
        np.cumsum([len(k) for k in kp_pairs], out=ptr[1:])
    parts = [np.asarray(k, np.int32).reshape(-1, 4) for k in kp_pairs if len(k)]
    flat = np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros((0, 4), np.int32)
    return flat, ptr


def _xy_block(xy_frames, n_frames):
    """The frames' coordinates as one float32[F, 2, n_max, 2] block (what the hot path hands over already is one)."""
    if isinstance(xy_frames, np.ndarray) and xy_frames.ndim == 4 and xy_frames.dtype == np.float32 and len(xy_frames) >= n_frames:
        return np.ascontiguousarray(xy_frames[:n_frames])
    per = [np.asarray(xy_frames[f], np.float32) for f in range(n_frames)]
    n_max = max([max(len(p[0]), len(p[1])) for p in per] + [1])
    out = np.zeros((n_frames, 2, n_max, 2), np.float32)
    for f, p in enumerate(per):
        out[f, 0, :len(p[0])] = p[0]
        out[f, 1, :len(p[1])] = p[1]
    return out


def chain_tracks(kp_pairs, xy_frames, flat=None) -> TrackTable:
    """kp_pairs[i]: int[n_i, 4] supporters of pair (i, i+1) as key-point indices (L_i, R_i, L_{i+1}, R_{i+1}), in the
    order the reference iterates them.  xy_frames[f]: float32[2, n, 2] key-point coordinates (left, right) of frame f
    (or the whole movie's float32[F, 2, n_max, 2]).  flat: (int32[M, 4], int64[P + 1]) if the caller already holds the
    supporters that way.  Follows DataBase.find_tracks_between_consecutive_frame (DataBase.py:102-153); runs in the
    library (slam_chain_tracks)."""
    from . import _lib
    P = len(kp_pairs)
    kp, ptr = flat if flat is not None else flatten_pairs(kp_pairs)
    kp = np.ascontiguousarray(kp, np.int32).reshape(-1, 4)
    ptr = np.ascontiguousarray(ptr, np.int64)
    if len(ptr) != P + 1 or ptr[-1] != len(kp):
        raise ValueError("pair pointer does not match the supporters")
    M = len(kp)
    xy = _xy_block(xy_frames, P + 1)
    n_max = xy.shape[2]
    n_kp = int(max(int(kp[:, 0].max()), int(kp[:, 2].max())) + 1) if M else 0
    if M and (n_kp > n_max or int(kp[:, 1].max()) >= n_max or int(kp[:, 3].max()) >= n_max or int(kp.min()) < 0):
        raise IndexError("supporter key-point index outside the frames' coordinate arrays")
    track_ptr = np.empty(M + 1, np.int64)
    obs_frame, obs_kp, obs_xy = np.empty(2 * M, np.int32), np.empty((2 * M, 2), np.int32), np.empty((2 * M, 4), np.float64)
    frame_ptr, frame_tracks = np.empty(P + 2, np.int64), np.empty(2 * M, np.int64)
    n_out = np.zeros(2, np.int64)
    code = _lib.lib().slam_chain_tracks(kp.ctypes.data, ptr.ctypes.data, P, n_kp, xy.ctypes.data, n_max, track_ptr.ctypes.data,
                                        obs_frame.ctypes.data, obs_kp.ctypes.data, obs_xy.ctypes.data, frame_ptr.ctypes.data,
                                        frame_tracks.ctypes.data, n_out[0:].ctypes.data, n_out[1:].ctypes.data)
    if code == _lib.SLAM_E_DUPLICATE:
        raise ValueError("duplicate key-points inside a pair: use the object DataBase")
    _lib.check(code)
    T, O = int(n_out[0]), int(n_out[1])
    return TrackTable(P + 1, T, track_ptr[:T + 1], obs_frame[:O], obs_kp[:O], obs_xy[:O], frame_ptr, frame_tracks[:O])


def chain_tracks_numpy(kp_pairs, xy_frames) -> TrackTable:
    """The same rule with one searchsorted per pair in numpy (the first implementation; what chain_tracks is tested
    against)."""
    P = len(kp_pairs)
    if not pairs_are_chainable(kp_pairs):
        raise ValueError("duplicate key-points inside a pair: use the object DataBase")
    n_tracks = 0
    last_kp = np.empty(0, np.int64)        # sorted frame-i left key-points that carry a track ...
    last_tid = np.empty(0, np.int64)       # ... and the track they carry
    o_track, o_frame, o_kp, o_xy = [], [], [], []            # observations in creation order
    f_lists = [[] for _ in range(P + 1)]
    for i, kp in enumerate(kp_pairs):
        kp = np.asarray(kp, np.int64).reshape(-1, 4)
        n = len(kp)
        k0, k1 = kp[:, 0], kp[:, 2]
        pos = np.searchsorted(last_kp, k0)
        hit = np.zeros(n, bool)
        inb = pos < len(last_kp)
        hit[inb] = last_kp[pos[inb]] == k0[inb]
        tid = np.empty(n, np.int64)
        tid[hit] = last_tid[pos[hit]]
        new = ~hit
        n_new = int(new.sum())
        tid[new] = n_tracks + np.arange(n_new)
        n_tracks += n_new
        # new tracks observe frame i (features of the pair's first half) ...
        o_track.append(tid[new]); o_frame.append(np.full(n_new, i, np.int32)); o_kp.append(kp[new][:, :2])
        o_xy.append(_coords(xy_frames[i], kp[new][:, 0], kp[new][:, 1]))
        # ... and every supporter observes frame i+1
        o_track.append(tid); o_frame.append(np.full(n, i + 1, np.int32)); o_kp.append(kp[:, 2:])
        o_xy.append(_coords(xy_frames[i + 1], kp[:, 2], kp[:, 3]))
        f_lists[i].append(tid[new])
        f_lists[i + 1].append(tid)
        order = np.argsort(k1, kind="stable")
        last_kp, last_tid = k1[order], tid[order]
    if o_track:
        o_track, o_frame = np.concatenate(o_track), np.concatenate(o_frame)
        o_kp, xy = np.concatenate(o_kp).reshape(-1, 2), np.concatenate(o_xy).reshape(-1, 4)
    else:
        o_track, o_frame = np.empty(0, np.int64), np.empty(0, np.int32)
        o_kp, xy = np.empty((0, 2), np.int64), np.empty((0, 4), np.float64)
    order = np.lexsort((o_frame, o_track))
    o_track, o_frame, o_kp, xy = o_track[order], o_frame[order], o_kp[order], xy[order]
    track_ptr = np.zeros(n_tracks + 1, np.int64)
    np.cumsum(np.bincount(o_track, minlength=n_tracks), out=track_ptr[1:])
    frame_ptr = np.zeros(P + 2, np.int64)
    ft = []
    for f in range(P + 1):
        a = np.concatenate(f_lists[f]) if f_lists[f] else np.empty(0, np.int64)
        ft.append(a)
        frame_ptr[f + 1] = frame_ptr[f] + len(a)
    frame_tracks = np.concatenate(ft) if ft else np.empty(0, np.int64)
    return TrackTable(P + 1, n_tracks, track_ptr, o_frame.astype(np.int32), o_kp.astype(np.int32), xy, frame_ptr, frame_tracks)


def supporters_flat(stats, mask, tracked_kp):
    """Supporter key-point indices of a whole ShardResult (pipeline.py) in one piece: stats int32[P,8] (column 0 =
    tracked count), mask bool[P,n_max], tracked_kp int32[P,n_max,4] (host copy) -> (int32[M,4], int64[P+1] row pointer).
    Order inside a pair is the tracked order, ascending left key-point of frame i, which is the order the reference's
    features arrive in."""
    P, n_max = mask.shape[0], mask.shape[1]
    valid = np.asarray(mask, bool) & (np.arange(n_max)[None, :] < np.asarray(stats)[:, :1])
    ptr = np.zeros(P + 1, np.int64)
    np.cumsum(valid.sum(1), out=ptr[1:])
    kp = np.ascontiguousarray(tracked_kp, np.int32).reshape(P, n_max, 4)
    rows = kp.view(np.dtype([("kp", np.int32, 4)])).reshape(P, n_max)        # 16-byte items: one gather, not four
    return rows[valid].view(np.int32).reshape(-1, 4), ptr


def supporters_from_shard(stats, mask, tracked_kp):
    """Per-pair list of the above (views into the one array)."""
    flat, ptr = supporters_flat(stats, mask, tracked_kp)
    return [flat[ptr[p]:ptr[p + 1]] for p in range(len(ptr) - 1)]

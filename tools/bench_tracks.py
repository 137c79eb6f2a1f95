"""Host-side timing of track-database construction: the object DataBase (the reference's
pair-by-pair Python chaining, slam_amd/types.py:DataBase) against the array-backed chaining (slam_amd/tracks.py)
on synthetic supporter index arrays of the whole-movie shape.  CPU only, single thread.

    python tools/bench_tracks.py [n_frames] [supporters_per_pair]
"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from slam_amd import tracks  # noqa: E402
from slam_amd.types import ArrayDataBase, DataBase, Feature  # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 3450
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1166
    n_kp = 2000
    rng = np.random.default_rng(0)
    xy = [rng.random((2, n_kp, 2)).astype(np.float32) * 1000 for _ in range(F)]
    kp_pairs, prev_l1 = [], np.empty(0, np.int64)
    for i in range(F - 1):
        keep = prev_l1[rng.random(len(prev_l1)) < 0.85][:S]                # ~85 % of the tracks continue
        rest = np.setdiff1d(np.arange(n_kp), keep)
        l0 = np.sort(np.concatenate([keep, rng.choice(rest, S - len(keep), replace=False)]))
        l1 = rng.choice(n_kp, S, replace=False)
        kp_pairs.append(np.stack([l0, rng.integers(0, n_kp, S), l1, rng.integers(0, n_kp, S)], 1).astype(np.int32))
        prev_l1 = l1
    glob = [np.hstack([np.eye(3), np.zeros((3, 1))])] * F
    percs = [90.0] * (F - 1)

    t0 = time.perf_counter()
    table = tracks.chain_tracks(kp_pairs, xy)
    adb = ArrayDataBase(table, kp_pairs, xy, percs, glob, glob)
    t_arr = time.perf_counter() - t0

    t0 = time.perf_counter()
    feats = []
    for i, kp in enumerate(kp_pairs):
        pair = []
        for f, c in ((i, (0, 1)), (i + 1, (2, 3))):
            xl, xr = xy[f][0], xy[f][1]
            a = np.empty(len(kp), dtype=object)
            for j in range(len(kp)):
                l, r = int(kp[j, c[0]]), int(kp[j, c[1]])
                a[j] = Feature(l, r, float(xl[l, 0]), float(xr[r, 0]), float(xl[l, 1]), float(xr[r, 1]))
            pair.append(a)
        feats.append(pair)
    t_feat = time.perf_counter() - t0
    t0 = time.perf_counter()
    db = DataBase(feats, percs, glob, glob)
    t_obj = time.perf_counter() - t0
    same = (db.get_num_tracks() == adb.get_num_tracks()
            and np.array_equal(table.track_len(), [t.get_track_len() for t in db.get_tracks()])
            and all(np.array_equal(a.get_tracks_ids(), b.get_tracks_ids()) for a, b in zip(db.get_frames(), adb.get_frames())))
    print(f"frames {F}  supporters/pair {S}  tracks {table.n_tracks}  observations {len(table.obs_frame)}  identical {same}")
    print(f"array-backed chain + ArrayDataBase : {t_arr:8.3f} s")
    print(f"Feature objects for the pair lists  : {t_feat:8.3f} s")
    print(f"object DataBase (reference rule)    : {t_obj:8.3f} s")


if __name__ == "__main__":
    main()

"""Generates tests/golden/*.npz by running the REFERENCE's own Python functions.

Run in the build container only (needs /root/reference; the GPU box never has it):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

`utils/utills.py` imports cv2, gtsam and DataDirectory.Data (which loads KITTI from
the author's disk) at import time, none of which exist offline.  Following SURVEY
section 8(c) they are replaced in `sys.modules` by in-memory stand-ins BEFORE the
import:
  * cv2: BFMatcher / solvePnP / Rodrigues bound to the CPU oracle's restatement of
    the OpenCV algorithms (oracle/slam_oracle.py) -- results that flow through them
    are "parity unpinned"; everything else the reference computes in Python/numpy
    (rect test, 4-view join, DLT triangulation via np.linalg.svd, projection,
    consensus test, adaptive RANSAC control flow, pose composition, DataBase
    track chaining) is the reference's OWN arithmetic and pins the oracle.
  * gtsam, gtsam.utils.plot: MagicMock (never called on this path).
  * DataDirectory.Data: a stub whose KITTI answers get_K/get_M1/get_M2 and serves
    the synthetic detections through get_image + ALG.detectAndCompute.
Only inputs and outputs (data) are stored; no reference source text.
"""
import os
import sys
import types
from unittest import mock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import slam_oracle as orc  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

PNP_LOG = []           # every stand-in solvePnP call: (flag, object points, ok)


def install_standins(seq_holder):
    cv2 = types.ModuleType("cv2")
    cv2.NORM_HAMMING, cv2.NORM_HAMMING2 = orc.NORM_HAMMING, orc.NORM_HAMMING2
    cv2.SOLVEPNP_ITERATIVE, cv2.SOLVEPNP_AP3P = orc.SOLVEPNP_ITERATIVE, orc.SOLVEPNP_AP3P

    class _Alg:
        def detectAndCompute(self, img, _mask):
            f, side = img                       # the "image" is a (frame index, side) token
            fr = seq_holder["frames"][f]
            n = int(fr["count"][side])
            kpts = [orc.KeyPoint(x, y) for x, y in fr["xy"][side, :n]]
            return kpts, fr["desc"][side, :n, :61].copy()

    cv2.AKAZE_create = lambda: _Alg()

    class BFMatcher:
        def __init__(self, norm, crossCheck=False):
            self.norm, self.cross = norm, crossCheck

        def match(self, q, t):
            assert self.cross
            return list(orc.to_dmatches(*orc.bf_match(q, t, self.norm)))

        def knnMatch(self, q, t, k):
            idx, dist = orc.knn2_from_matrix(orc.hamming_matrix(q, t, self.norm))
            return [[orc.DMatch(i, idx[i, 0], dist[i, 0]), orc.DMatch(i, idx[i, 1], dist[i, 1])]
                    for i in range(len(q))]

    cv2.BFMatcher = BFMatcher

    def solvePnP(obj, img, K, dist, flags=0):
        if flags == orc.SOLVEPNP_AP3P:
            ok, r, t = orc.solve_pnp_ap3p(obj, img, K)
        else:
            ok, r, t = orc.solve_pnp_iterative(obj, img, K)
        PNP_LOG.append((flags, np.array(obj), bool(ok)))
        return ok, r, t

    cv2.solvePnP = solvePnP
    cv2.Rodrigues = lambda r: (orc.rodrigues_vec_to_mat(np.asarray(r).reshape(3)), None)
    sys.modules["cv2"] = cv2

    for name in ("gtsam", "gtsam.utils", "gtsam.utils.plot"):
        sys.modules[name] = mock.MagicMock()

    class _Kitti:
        def get_K(self): return orc.K.copy()
        def get_M1(self): return orc.M1.copy()
        def get_M2(self): return orc.M2.copy()
        def get_image(self, i): return (i, 0), (i, 1)

    data = types.ModuleType("DataDirectory.Data")
    data.KITTI = _Kitti()
    pkg = types.ModuleType("DataDirectory")
    pkg.__path__ = [os.path.join(REF, "DataDirectory")]
    pkg.Data = data
    sys.modules["DataDirectory"] = pkg
    sys.modules["DataDirectory.Data"] = data
    sys.path.insert(0, REF)


def feats_to_arrays(feats):
    kp = np.array([[f.get_left_kpt(), f.get_right_kpt()] for f in feats], np.int32).reshape(-1, 2)
    xy = np.array([[*f.get_left_coor(), *f.get_right_coor()] for f in feats], np.float64).reshape(-1, 4)
    return kp, xy


def main():
    holder = {}
    install_standins(holder)
    from utils import utills as U          # THE REFERENCE
    from DataBaseDirectory import DataBase as RDB

    cfg = SynthConfig(n_frames=40, n_features=320, seed=7, spawn_per_frame=40, ragged=13)
    seq = SynthSequence(cfg)
    frames = [seq.frame(i) for i in range(8)]
    holder["frames"] = frames
    K, M1, M2 = orc.K, orc.M1, orc.M2
    rng = np.random.default_rng(2024)
    g = {}

    # ---- rows 3/4/5/11 on one synthetic pair, through the reference's own driver -----------------
    s0 = U.read_and_rec_match(2)
    s1 = U.read_and_rec_match(3)
    l0k, l0d, r0k, m0, rec0 = s0
    l1k, l1d, r1k, m1, rec1 = s1
    g["pair_rec0"] = np.array(rec0, np.int64)
    g["pair_rec1"] = np.array(rec1, np.int64)
    g["pair_m0"] = np.array([[m.queryIdx, m.trainIdx, m.distance] for m in m0], np.int64)
    g["pair_m1"] = np.array([[m.queryIdx, m.trainIdx, m.distance] for m in m1], np.int64)
    tm = U.match(l0d, l1d)
    g["pair_tm"] = np.array([[m.queryIdx, m.trainIdx, m.distance] for m in tm], np.int64)
    a, b, c = U.find_kpts_in_all_4_rec(tm, U.create_rec_dic(m0, rec0), U.create_rec_dic(m1, rec1))
    g["pair_join"] = np.array([a, b, c], np.int64)
    np.random.seed(4242)
    PNP_LOG.clear()
    T, f0, f1, perc, mask = U.find_features_in_consecutive_frames(l0k, l0d, r0k, m0, rec0, l1k, l1d, r1k, m1, rec1)
    g["pair_T"] = T
    g["pair_perc"] = np.float64(perc)
    g["pair_mask"] = np.asarray(mask, bool)
    g["pair_f0_kp"], g["pair_f0_xy"] = feats_to_arrays(f0)
    g["pair_f1_kp"], g["pair_f1_xy"] = feats_to_arrays(f1)
    g["pair_n_ap3p_calls"] = np.int64(sum(1 for f, _, _ in PNP_LOG if f == orc.SOLVEPNP_AP3P))
    g["pair_X"] = U.triangulate(K @ M1, K @ M2, g["pair_f0_xy"][:, :2], g["pair_f0_xy"][:, 2:])

    # ---- row 6: triangulation (reference numpy + LAPACK) --------------------------------------
    n = 400
    Xw = np.stack([rng.uniform(-25, 25, n), rng.uniform(-4, 3, n), rng.uniform(3, 120, n)], 1)
    xl = orc.project(Xw, K @ M1) + rng.normal(0, 0.4, (n, 2))
    xr = orc.project(Xw, K @ M2) + rng.normal(0, 0.4, (n, 2))
    xl = xl.astype(np.float32).astype(np.float64)
    xr = xr.astype(np.float32).astype(np.float64)
    xr[:5] = xl[:5] + rng.normal(0, 2.0, (5, 2))           # near-zero / negative disparity
    xr[5:10, 0] = xl[5:10, 0] + 3.0                        # behind the rig
    g["tri_xl"], g["tri_xr"] = xl, xr
    g["tri_X"] = U.triangulate(K @ M1, K @ M2, xl, xr)

    # ---- row 8: projection / consensus incl. the 4-px quirk, z<=0 and NaN poses -----------------
    n = 600
    Xc = np.stack([rng.uniform(-20, 20, n), rng.uniform(-4, 3, n), rng.uniform(2, 90, n)], 1)
    Tt = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.01, -0.03, 0.005])), np.array([[0.05], [-0.02], [-0.8]])])
    Tr = U.compose_transformations(Tt, M2)
    l1 = U.project(Xc, K @ Tt)
    r1 = U.project(Xc, K @ Tr)
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = rng.uniform(0, 8, n)
    rad[:40] = np.array([3.9, 4.1, 3.999999, 4.000001] * 10)
    l1o = l1 + np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
    r1o = r1 + rng.normal(0, 1.5, (n, 2))
    l1o = l1o.astype(np.float32).astype(np.float64)
    r1o = r1o.astype(np.float32).astype(np.float64)
    Xc[40:44, 2] = [-5.0, 0.0, -0.5, 1e-9]                  # behind / on the camera plane
    g["cons_X"], g["cons_T"], g["cons_l1"], g["cons_r1"] = Xc, Tt, l1o, r1o
    with np.errstate(all="ignore"):
        g["cons_mask"] = U.consensus_match(K, Tt, M2, Xc, l1o, r1o, 2)
        g["cons_proj_l"] = U.project(Xc, K @ Tt)
        g["cons_compose"] = Tr
        Tnan = np.hstack([np.eye(3), np.full((3, 1), np.nan)])
        g["cons_mask_nan"] = U.consensus_match(K, Tnan, M2, Xc, l1o, r1o, 2)
    # many hypotheses against the same points: counts are what the scoring kernel must reproduce
    H = 48
    Ts = []
    for h in range(H):
        dr = rng.normal(0, 0.004, 3)
        dt = rng.normal(0, 0.03, 3)
        Ts.append(np.hstack([orc.rodrigues_vec_to_mat(np.array([0.01, -0.03, 0.005]) + dr),
                             (Tt[:, 3] + dt)[:, None]]))
    Ts = np.array(Ts)
    g["score_T"] = Ts
    g["score_masks"] = np.array([U.consensus_match(K, Ts[h], M2, Xc, l1o, r1o, 2) for h in range(H)])

    # ---- row 9/10: the adaptive RANSAC loop, seeded ------------------------------------------
    for tag, seed, frac_out in (("a", 99, 0.1), ("b", 100, 0.45)):
        n = 260
        Xr = np.stack([rng.uniform(-20, 20, n), rng.uniform(-4, 3, n), rng.uniform(4, 70, n)], 1)
        l = U.project(Xr, K @ Tt) + rng.normal(0, 0.3, (n, 2))
        r = U.project(Xr, K @ Tr) + rng.normal(0, 0.3, (n, 2))
        bad = rng.random(n) < frac_out
        l[bad] += rng.normal(0, 30, (int(bad.sum()), 2))
        l = l.astype(np.float32).astype(np.float64)
        r = r.astype(np.float32).astype(np.float64)
        np.random.seed(seed)
        PNP_LOG.clear()
        T, m = U.online_est_pnp_ransac(4, Xr, M1, l, M2, r, l, M2, r, K, acc=2)
        st = np.random.get_state()
        g[f"ransac_{tag}_X"], g[f"ransac_{tag}_l1"], g[f"ransac_{tag}_r1"] = Xr, l, r
        g[f"ransac_{tag}_seed"] = np.int64(seed)
        g[f"ransac_{tag}_T"], g[f"ransac_{tag}_mask"] = T, np.asarray(m, bool)
        g[f"ransac_{tag}_draws"] = np.int64(sum(1 for f, _, _ in PNP_LOG if f == orc.SOLVEPNP_AP3P))
        g[f"ransac_{tag}_fails"] = np.int64(sum(1 for f, _, ok in PNP_LOG if f == orc.SOLVEPNP_AP3P and not ok))
        g[f"ransac_{tag}_next_rand"] = np.int64(np.random.randint(0, 1 << 30))   # stream position check
    # (the reference raises on an empty set)
    try:
        U.online_est_pnp_ransac(4, np.zeros((0, 3)), M1, np.zeros((0, 2)), M2, np.zeros((0, 2)), np.zeros((0, 2)), M2,
                                np.zeros((0, 2)), K, acc=2)
        g["ransac_empty_raises"] = np.array("none")
    except Exception as e:                                    # noqa: BLE001
        g["ransac_empty_raises"] = np.array(type(e).__name__)

    # ---- row 11/12 + N1: whole-movie driver on 8 frames and the DataBase built from it -----------------
    U.MOVIE_LEN = len(frames)
    np.random.seed(31337)
    feats, percs, glob, rel = U.find_features_in_consecutive_frames_whole_movie()
    g["movie_rel"] = np.array(rel)
    g["movie_glob"] = np.array(glob)
    g["movie_perc"] = np.array(percs)
    for i, (a0, a1) in enumerate(feats):
        g[f"movie_f0_kp_{i}"], g[f"movie_f0_xy_{i}"] = feats_to_arrays(a0)
        g[f"movie_f1_kp_{i}"], g[f"movie_f1_xy_{i}"] = feats_to_arrays(a1)
    db = RDB.DataBase(feats, percs, glob, rel)
    g["db_num_tracks"] = np.int64(db.get_num_tracks())
    g["db_track_len"] = np.array([t.get_track_len() for t in db.get_tracks()], np.int64)
    g["db_track_first"] = np.array([t.get_first_frame_ind() for t in db.get_tracks()], np.int64)
    g["db_frame_ntracks"] = np.array([f.get_tracks_num() for f in db.get_frames()], np.int64)
    for i, f in enumerate(db.get_frames()):
        g[f"db_frame_tracks_{i}"] = f.get_tracks_ids()
    g["db_poses"] = db.initial_estimate_poses()
    g["traj_centres"] = U.left_cameras_trajectory(glob)

    # inputs of the movie (so tests need not regenerate them with a possibly changed generator)
    g["movie_desc"] = np.array([f["desc"] for f in frames])
    g["movie_xy"] = np.array([f["xy"] for f in frames])
    g["movie_count"] = np.array([f["count"] for f in frames])

    np.savez_compressed(os.path.join(OUT, "reference_vectors.npz"), **g)
    print("wrote", len(g), "arrays;", "pair T=", g["pair_T"].shape, "tracked", len(g["pair_mask"]),
          "supporters", int(g["pair_mask"].sum()), "ransac draws", int(g["ransac_a_draws"]), int(g["ransac_b_draws"]),
          "fails", int(g["ransac_a_fails"]), int(g["ransac_b_fails"]), "tracks", int(g["db_num_tracks"]))


if __name__ == "__main__":
    main()

"""Second batch of vectors recorded from the REFERENCE's own functions -> tests/golden/reference_vectors_more.npz.

Run in the build container only (needs /root/reference), AFTER make_golden.py (it reuses that script's stand-ins
for cv2 / gtsam / DataDirectory.Data and the 8-frame movie stored in reference_vectors.npz):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_more.py

Covers the entry points the first batch leaves out:
  * match(..., REGULAR_KNN) + significance_test  (utils/utills.py:105-119, :131-152): the ratio test is the
    reference's own float64 comparison, recorded on the movie and on a table of (d1, d2) around 0.6 * d2;
  * compute_trans_between_cur_to_next  (:259-293, the get_matches_coor path) on a seeded pair;
  * apply_full_consensus_match  (:1103-1115) on a non-consecutive pair;
  * find_loop_cand_by_consensus + create_little_tracks  (:1118-1167): candidates that pass and that fail the
    strict inlier-percentage threshold, all drawing from ONE global random stream in candidate order;
  * online_est_pnp_ransac on a set with fewer than 6 supporters (:558-559): pose None, mask still returned.
    (A set on which EVERY solve fails is not recorded: the reference's retry loop, :506-507, never returns on it.)
Only inputs and outputs (data) are stored.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import make_golden as MG  # noqa: E402  (stand-ins; also puts the repo root and oracle/ on sys.path)

orc = MG.orc


def triples(ms):
    return np.array([[m.queryIdx, m.trainIdx, m.distance] for m in ms], np.int64).reshape(-1, 3)


def main():
    base = np.load(os.path.join(HERE, "reference_vectors.npz"), allow_pickle=False)
    frames = [dict(desc=base["movie_desc"][i], xy=base["movie_xy"][i], count=base["movie_count"][i])
              for i in range(len(base["movie_count"]))]
    holder = {"frames": frames}
    MG.install_standins(holder)
    from utils import utills as U          # THE REFERENCE
    K, M1, M2 = orc.K, orc.M1, orc.M2
    rng = np.random.default_rng(4096)
    g = {}

    # ---- row 2: kNN(2) + ratio test ------------------------------------------------------------
    s2, s3 = U.read_and_rec_match(2), U.read_and_rec_match(3)
    g["knn_temporal"] = triples(U.match(s2[1], s3[1], U.REGULAR_KNN))
    n1 = int(base["movie_count"][2][1])
    g["knn_stereo"] = triples(U.match(s2[1], base["movie_desc"][2, 1, :n1, :61].copy(), U.REGULAR_KNN))
    d2 = np.repeat(np.arange(0, 489), 5)                                   # HAMMING on 61 bytes reaches 488
    d1 = np.floor(0.6 * d2).astype(np.int64) + np.tile(np.arange(-2, 3), 489)
    d1 = np.clip(d1, 0, None)
    pairs = [[orc.DMatch(i, 0, int(a)), orc.DMatch(i, 1, int(b))] for i, (a, b) in enumerate(zip(d1, d2))]
    ok, _ = U.significance_test(pairs, U.RATIO)
    passed = np.zeros(len(pairs), bool)
    passed[[m.queryIdx for m in ok]] = True
    g["sig_d1"], g["sig_d2"], g["sig_pass"] = d1, d2, passed

    # ---- row 11 (older API): compute_trans_between_cur_to_next ---------------------------------
    np.random.seed(777)
    T, mask = U.compute_trans_between_cur_to_next(*s2, *s3)
    g["cur_next_T"], g["cur_next_mask"] = T, np.asarray(mask, bool)
    g["cur_next_next_rand"] = np.int64(np.random.randint(0, 1 << 30))

    # ---- row 12: apply_full_consensus_match on a non-consecutive pair --------------------------
    np.random.seed(778)
    T, f0, f1, sup, perc = U.apply_full_consensus_match(1, 4)
    g["full_T"], g["full_mask"], g["full_perc"] = T, np.asarray(sup, bool), np.float64(perc)
    g["full_f0_kp"], g["full_f0_xy"] = MG.feats_to_arrays(f0)
    g["full_f1_kp"], g["full_f1_xy"] = MG.feats_to_arrays(f1)
    g["full_next_rand"] = np.int64(np.random.randint(0, 1 << 30))

    # ---- row 12: find_loop_cand_by_consensus ---------------------------------------------------
    # inlier percentage of every candidate first (fresh stream each), to place thresholds between them
    cur, cands, pg = 4, [3, 0, 6, 2], [30, 0, 60, 20]
    percs = []
    for c in cands:
        np.random.seed(900 + c)
        percs.append(float(U.apply_full_consensus_match(c, cur)[4]))
    g["loop_probe_perc"] = np.array(percs)
    srt = np.sort(percs)
    thresholds = [0.0, float((srt[0] + srt[1]) / 2), float((srt[-2] + srt[-1]) / 2), 100.0]
    g["loop_cur"], g["loop_cands"], g["loop_pg"] = np.int64(cur), np.array(cands), np.array(pg)
    g["loop_thresholds"] = np.array(thresholds)
    for k, thr in enumerate(thresholds):
        np.random.seed(779)
        with contextlib.redirect_stdout(io.StringIO()) as out:          # the reference prints "pg: perc" per candidate
            res = U.find_loop_cand_by_consensus(cands, pg, cur, thr)
        g[f"loop_{k}_printed_perc"] = np.array([float(line.split(":")[1]) for line in out.getvalue().splitlines() if ":" in line])
        g[f"loop_{k}_passed_pg"] = np.array([r[0] for r in res], np.int64)
        g[f"loop_{k}_next_rand"] = np.int64(np.random.randint(0, 1 << 30))
        for j, (_, tracks) in enumerate(res):
            fr = np.array([t.get_frames_ids() for t in tracks], np.int64).reshape(-1, 2)
            kp = np.array([[t.get_feature_at_frame(int(f)).get_left_kpt() for f in t.get_frames_ids()] +
                           [t.get_feature_at_frame(int(f)).get_right_kpt() for f in t.get_frames_ids()]
                           for t in tracks], np.int64).reshape(-1, 4)       # (L_first, L_second, R_first, R_second)
            g[f"loop_{k}_{j}_frames"], g[f"loop_{k}_{j}_kp"] = fr, kp
            g[f"loop_{k}_{j}_ids"] = np.array([t.get_id() for t in tracks], np.int64)

    # ---- row 10: fewer than 6 supporters -> pose None, RANSAC mask still returned (:526-530, :558-559) ------------
    n = 24
    Xn = np.stack([rng.uniform(-20, 20, n), rng.uniform(-4, 3, n), rng.uniform(4, 70, n)], 1)
    ln = np.stack([rng.uniform(0, 1241, n), rng.uniform(0, 376, n)], 1).astype(np.float32).astype(np.float64)
    rn = np.stack([rng.uniform(0, 1241, n), rng.uniform(0, 376, n)], 1).astype(np.float32).astype(np.float64)
    # five points that do agree with one pose: the best hypothesis collects 1..5 supporters, still fewer than 6
    Tt = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.01, -0.03, 0.005])), np.array([[0.05], [-0.02], [-0.8]])])
    ln[:5] = (U.project(Xn[:5], K @ Tt) + rng.normal(0, 0.2, (5, 2))).astype(np.float32)
    rn[:5] = (U.project(Xn[:5], K @ U.compose_transformations(Tt, M2)) + rng.normal(0, 0.2, (5, 2))).astype(np.float32)
    np.random.seed(780)
    MG.PNP_LOG.clear()
    with np.errstate(all="ignore"):
        T, m = U.online_est_pnp_ransac(4, Xn, M1, ln, M2, rn, ln, M2, rn, K, acc=2)
    g["few_X"], g["few_l1"], g["few_r1"] = Xn, ln, rn
    g["few_T_is_none"] = np.bool_(T is None)
    g["few_mask"] = np.asarray(m, bool)
    g["few_draws"] = np.int64(sum(1 for f, _, _ in MG.PNP_LOG if f == orc.SOLVEPNP_AP3P))
    g["few_fails"] = np.int64(sum(1 for f, _, ok in MG.PNP_LOG if f == orc.SOLVEPNP_AP3P and not ok))
    g["few_next_rand"] = np.int64(np.random.randint(0, 1 << 30))

    np.savez_compressed(os.path.join(HERE, "reference_vectors_more.npz"), **g)
    print("wrote", len(g), "arrays; knn", len(g["knn_temporal"]), len(g["knn_stereo"]), "sig pass", int(passed.sum()),
          "loop percs", np.round(percs, 2), "thresholds", np.round(thresholds, 2),
          "passed", [list(g[f"loop_{k}_passed_pg"]) for k in range(len(thresholds))],
          "few: T none", bool(g["few_T_is_none"]), "supp", int(g["few_mask"].sum()), "draws", int(g["few_draws"]),
          "fails", int(g["few_fails"]))


if __name__ == "__main__":
    main()

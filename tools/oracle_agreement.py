"""How often the whole pipeline agrees with the CPU oracle, pair by pair (development / reporting tool; imports oracle/
as the checker).  Prints the fraction of pairs with identical tracked sets, supporter masks, iteration counts and
winning draws, the worst pose deviation, and for the pairs that differ how far the triangulated points are apart.

    python tools/oracle_agreement.py [n_frames] [n_features]
    python tools/oracle_agreement.py --bench [--workers 16]     the bench workload itself (3450 frames x 2000 features,
                                                                seed 1234), oracle pairs spread over CPU worker processes
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import c_oracle  # noqa: E402
import slam_oracle as orc  # noqa: E402
from slam_amd import ops  # noqa: E402
from slam_amd.pipeline import ShardTracker, consecutive_pairs  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence  # noqa: E402


_G = {}


def _oracle_pair(p):
    """Worker: the oracle's result for pair p, condensed to what the comparison needs."""
    desc, xy, count = _G["data"]
    fr = lambda i: dict(dsc_l=desc[i, 0, :count[i, 0], :61], dsc_r=desc[i, 1, :count[i, 1], :61],
                        xy_l=xy[i, 0, :count[i, 0]], xy_r=xy[i, 1, :count[i, 1]])
    np.random.seed(1000 + p)
    st = {}
    with np.errstate(all="ignore"):
        w = orc.track_pair(fr(p), fr(p + 1), stats=st)
    # draws whose fourth index repeats one of the first three: AP3P picks its root by the fourth point's reprojection
    # error, which is rounding noise for a point the roots were computed from (tools/diverging_pairs.py)
    np.random.seed(1000 + p)
    n, noisy = len(w["kp"]), 0
    for _ in range(st["draws"] + 64):            # + slack: the other side may have consumed a few more draws
        i = np.random.randint(0, n, 4)
        noisy += int(i[3] in i[:3])
    return p, w["kp"].astype(np.int32), np.packbits(w["mask"]), st["iterations"], st["best_draw"], w["T"], noisy


def bench_workload(workers):
    """Whole bench shard against the oracle.  The oracle pairs run first, in forked CPU workers, before this process
    touches the GPU (a forked child must not inherit an initialised HIP runtime)."""
    import json
    import multiprocessing as mp
    import time
    orc.bf_match = lambda x, y, norm=7: c_oracle.match_crosscheck_w64(c_oracle.pad64(x), c_oracle.pad64(y), norm)
    cfg = SynthConfig(n_frames=3450, n_features=2000, seed=1234)
    F = cfg.n_frames
    seq = SynthSequence(cfg)
    desc, xy, count = seq.frames(0, F)
    _G["data"] = (desc, xy, count)
    t0 = time.time()
    with mp.get_context("fork").Pool(workers) as pool:
        want = {}
        for k, r in enumerate(pool.imap_unordered(_oracle_pair, range(F - 1), chunksize=8)):
            want[r[0]] = r[1:]
            if k % 500 == 0:
                print(f"oracle pairs done: {k} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    t_cpu = time.time() - t0
    n_max = cfg.n_max
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    f0, f1 = consecutive_pairs(F)
    res = ShardTracker(F, F - 1, n_max, n_chunks=2).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                                        dev(count.reshape(2 * F)), f0, f1, 1000 + np.arange(F - 1))
    kp = res.tracked_kp.cpu().numpy()
    same = dict(tracked_sets=0, supporter_masks=0, iteration_counts=0, winning_draws=0)
    clean = dict(pairs=0, supporter_masks=0, iteration_counts=0, winning_draws=0, pose_rel_dev_max=0.0)
    pose_dev = np.zeros(F - 1)
    supp_diff, differing = [], []
    for p in range(F - 1):
        wkp, wmask, wit, wbest, wT, wnoisy = want[p]
        n = len(wkp)
        ok_kp = res.stats[p, 0] == n and np.array_equal(kp[p, :n], wkp)
        ok_mask = ok_kp and np.array_equal(np.packbits(res.mask[p, :n]), wmask)
        same["tracked_sets"] += int(ok_kp)
        same["supporter_masks"] += int(ok_mask)
        same["iteration_counts"] += int(res.stats[p, 2] == wit)
        same["winning_draws"] += int(res.stats[p, 4] == wbest)
        if wT is not None:
            pose_dev[p] = np.linalg.norm(res.pose[p] - wT) / np.linalg.norm(wT)
        if wnoisy == 0:                          # pairs whose sample stream holds no rounding-noise draw
            clean["pairs"] += 1
            clean["supporter_masks"] += int(ok_mask)
            clean["iteration_counts"] += int(res.stats[p, 2] == wit)
            clean["winning_draws"] += int(res.stats[p, 4] == wbest)
            clean["pose_rel_dev_max"] = max(clean["pose_rel_dev_max"], float(pose_dev[p]))
        if ok_kp and not ok_mask:
            supp_diff.append(int(np.unpackbits(np.packbits(res.mask[p, :n]) ^ wmask).sum()))
            differing.append(p)
    # north_star's third criterion: the trajectory error of the two runs against the generator's ground truth
    from slam_amd import trajectory
    from slam_amd.pipeline import compose_global
    gt = np.array([seq.gt_global(i) for i in range(F)])
    eye = np.eye(3, 4)[None]
    fill = lambda rel: np.where(np.isfinite(rel).all(axis=(1, 2), keepdims=True), rel, eye)
    rel_o = np.array([want[p][4] if want[p][4] is not None else np.full((3, 4), np.nan) for p in range(F - 1)])
    tr_g = trajectory.summary(compose_global(np.concatenate([eye, fill(res.pose)])), gt)
    tr_o = trajectory.summary(compose_global(np.concatenate([eye, fill(rel_o)])), gt)
    ate = {"ate_rmse_m_gpu": tr_g["ate_rmse_m"], "ate_rmse_m_oracle": tr_o["ate_rmse_m"],
           "ate_rel_diff": abs(tr_g["ate_rmse_m"] - tr_o["ate_rmse_m"]) / tr_o["ate_rmse_m"],
           "rpe_trans_pct_gpu": tr_g["rpe_trans_pct"], "rpe_trans_pct_oracle": tr_o["rpe_trans_pct"]}
    out = {"workload": "bench (3450 frames x 2000 features, seed 1234)", "pairs": F - 1, **same, **ate,
           "pose_rel_dev_median": float(np.median(pose_dev)), "pose_rel_dev_p99": float(np.quantile(pose_dev, 0.99)),
           "pose_rel_dev_max": float(pose_dev.max()), "pairs_with_pose_dev_above_1e-5": int((pose_dev > 1e-5).sum()),
           "supporters_differing_where_masks_differ": supp_diff, "pairs_where_masks_differ": differing,
           "pairs_without_a_repeated_fourth_sample_index": clean, "oracle_cpu_s": round(t_cpu, 1), "workers": workers}
    print(json.dumps(out))


def main():
    if "--bench" in sys.argv:
        w = int(sys.argv[sys.argv.index("--workers") + 1]) if "--workers" in sys.argv else 16
        return bench_workload(w)
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    orc.bf_match = lambda x, y, norm=7: c_oracle.match_crosscheck_w64(c_oracle.pad64(x), c_oracle.pad64(y), norm)
    cfg = SynthConfig(n_frames=F, n_features=N, seed=99, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    f0, f1 = consecutive_pairs(F)
    res = ShardTracker(F, F - 1, n_max, n_chunks=2).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                                        dev(count.reshape(2 * F)), f0, f1, 1000 + np.arange(F - 1))
    kp = res.tracked_kp.cpu().numpy()
    fr = lambda i: dict(dsc_l=desc[i, 0, :count[i, 0], :61], dsc_r=desc[i, 1, :count[i, 1], :61],
                        xy_l=xy[i, 0, :count[i, 0]], xy_r=xy[i, 1, :count[i, 1]])
    same_kp = same_mask = same_it = same_best = 0
    worst = 0.0
    notes = []
    P, Q = orc.K @ orc.M1, orc.K @ orc.M2
    for p in range(F - 1):
        np.random.seed(1000 + p)
        st = {}
        with np.errstate(all="ignore"):
            w = orc.track_pair(fr(p), fr(p + 1), stats=st)
        n = len(w["kp"])
        ok_kp = res.stats[p, 0] == n and np.array_equal(kp[p, :n], w["kp"])
        ok_mask = ok_kp and np.array_equal(res.mask[p, :n], w["mask"])
        same_kp += ok_kp
        same_mask += ok_mask
        same_it += res.stats[p, 2] == st["iterations"]
        same_best += res.stats[p, 4] == st["best_draw"]
        worst = max(worst, float(np.linalg.norm(res.pose[p] - w["T"]) / np.linalg.norm(w["T"])))
        if res.stats[p, 2] != st["iterations"] and len(notes) < 6 and ok_kp:
            k = w["kp"]
            xl = np.stack([xy[p, 0, k[:, 0]], xy[p, 1, k[:, 1]]], 1).astype(np.float64)      # [n, 2(l/r), 2]
            Xo = orc.triangulate(P, Q, xl[:, 0], xl[:, 1])
            Xg = ops.triangulate_f64(P, Q, dev(xl[:, 0]), dev(xl[:, 1])).cpu().numpy()
            rel = np.linalg.norm(Xg - Xo, axis=1) / np.linalg.norm(Xo, axis=1)
            j = int(np.argmax(rel))
            notes.append(f"pair {p}: iterations {int(res.stats[p, 2])} vs oracle {st['iterations']}; triangulated points: "
                         f"median rel diff {np.median(rel):.2e}, max {rel[j]:.2e} at a point of depth {Xo[j, 2]:.1f} m "
                         f"(disparity {xl[j, 0, 0] - xl[j, 1, 0]:.3f} px); points beyond 1e-9: {int((rel > 1e-9).sum())} of {n}")
    n = F - 1
    print(f"pairs {n}: tracked sets equal {same_kp}, supporter masks equal {same_mask}, iteration counts equal {same_it}, "
          f"winning draw equal {same_best}, worst relative pose deviation {worst:.3e}")
    for s in notes:
        print(s)


if __name__ == "__main__":
    main()

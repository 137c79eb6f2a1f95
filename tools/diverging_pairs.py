"""Where the pipeline and the oracle part ways on a given frame pair of the bench workload (reporting tool; imports
oracle/ as the checker).  For every draw of the pair's sample stream it solves and scores four ways -- solver
{oracle, GPU} x triangulated points {oracle (LAPACK SVD), GPU (Jacobi)} -- and prints the draws whose inlier counts
differ, with the pose deviation between the two solvers on the same points.

    python tools/diverging_pairs.py PAIR [PAIR ...]
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
from slam_amd.synth import SynthConfig, SynthSequence  # noqa: E402

N_DRAWS = 64


def main():
    orc.bf_match = lambda x, y, norm=7: c_oracle.match_crosscheck_w64(c_oracle.pad64(x), c_oracle.pad64(y), norm)
    seq = SynthSequence(SynthConfig(n_frames=3450, n_features=2000, seed=1234))
    P, Q = orc.K @ orc.M1, orc.K @ orc.M2
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for p in [int(a) for a in sys.argv[1:]]:
        fa, fb = seq.frame(p), seq.frame(p + 1)
        fr = lambda f: dict(dsc_l=f["desc"][0, :f["count"][0], :61], dsc_r=f["desc"][1, :f["count"][1], :61],
                            xy_l=f["xy"][0, :f["count"][0]], xy_r=f["xy"][1, :f["count"][1]])
        np.random.seed(1000 + p)
        st = {}
        with np.errstate(all="ignore"):
            w = orc.track_pair(fr(fa), fr(fb), stats=st)
        n = len(w["kp"])
        Xo = w["X"]
        Xg = ops.triangulate_f64(P, Q, dev(w["l0"]), dev(w["r0"])).cpu().numpy()
        relX = np.linalg.norm(Xg - Xo, axis=1) / np.linalg.norm(Xo, axis=1)
        np.random.seed(1000 + p)
        idx = np.stack([np.random.randint(0, n, 4) for _ in range(N_DRAWS)]).astype(np.int32)
        stride = (n + 63) // 64 * 64
        xy1 = np.zeros((1, stride, 4))
        xy1[0, :n, :2], xy1[0, :n, 2:] = w["l1"], w["r1"]
        cnt = torch.tensor([n], dtype=torch.int32).cuda()
        res = {}
        for tag, X in (("Xo", Xo), ("Xg", Xg)):
            Xp = np.zeros((1, stride, 3))
            Xp[0, :n] = X
            pose, status = ops.p3p_hypotheses(orc.K, dev(Xp), dev(xy1), cnt, dev(idx[None]))
            c, _ = ops.score_hypotheses(orc.K, orc.M2, dev(Xp), dev(xy1), cnt, pose, status)
            res["gpu", tag] = (pose.cpu().numpy()[0], status.cpu().numpy()[0], c.cpu().numpy()[0])
            po, so, co = np.full((N_DRAWS, 3, 4), np.nan), np.zeros(N_DRAWS, np.uint8), np.zeros(N_DRAWS, np.int32)
            with np.errstate(all="ignore"):
                for k in range(N_DRAWS):
                    T = orc.pnp(X[idx[k]], w["l1"][idx[k]], orc.K, orc.SOLVEPNP_AP3P)
                    if T is not None:
                        po[k], so[k] = T, 1
                        co[k] = int(np.sum(orc.consensus_match(orc.K, T, orc.M2, X, w["l1"], w["r1"], orc.SUPP_ERR)))
            res["orc", tag] = (po, so, co)
        print(f"pair {p}: T={n}, oracle iterations {st['iterations']}, best draw {st['best_draw']} ({st['best_count']}); "
              f"points: median rel diff {np.median(relX):.1e}, max {relX.max():.1e}")
        for k in range(N_DRAWS):
            c = {key: int(v[2][k]) for key, v in res.items()}
            s = {key: int(v[1][k]) for key, v in res.items()}
            if len(set(c.values())) > 1 or len(set(s.values())) > 1:
                dp = np.linalg.norm(res["gpu", "Xo"][0][k] - res["orc", "Xo"][0][k]) / np.linalg.norm(res["orc", "Xo"][0][k])
                dx = np.linalg.norm(res["orc", "Xg"][0][k] - res["orc", "Xo"][0][k]) / np.linalg.norm(res["orc", "Xo"][0][k])
                dup = len(set(idx[k].tolist())) < 4
                print(f"  draw {k:2d} idx {idx[k].tolist()}{' (repeated point)' if dup else ''}: counts orc/Xo {c['orc', 'Xo']} orc/Xg {c['orc', 'Xg']} "
                      f"gpu/Xo {c['gpu', 'Xo']} gpu/Xg {c['gpu', 'Xg']}; status {list(s.values())}; pose rel dev solver (same points) {dp:.1e}, "
                      f"points (same solver) {dx:.1e}")


if __name__ == "__main__":
    main()

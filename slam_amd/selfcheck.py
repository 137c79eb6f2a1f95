"""One small invocation of the HIP hot path on cuda:0, checked against the CPU oracle (used by
__graft_entry__.smoke(); the oracle is imported here only as the checker)."""
import numpy as np


def smoke():
    import torch

    import c_oracle
    import slam_oracle as orc
    from slam_amd.pipeline import ShardTracker, consecutive_pairs
    from slam_amd.synth import SynthConfig, SynthSequence

    torch.cuda.set_device(0)
    cfg = SynthConfig(n_frames=30, n_features=400, seed=21, spawn_per_frame=40, ragged=17)
    seq = SynthSequence(cfg)
    F = 3
    desc, xy, count = seq.frames(0, F)
    n_max = cfg.n_max
    trk = ShardTracker(F, F - 1, n_max)
    f0, f1 = consecutive_pairs(F)
    seeds = 1000 + np.arange(F - 1)
    res = trk.run(torch.from_numpy(desc.reshape(2 * F, n_max, 64)).cuda(), torch.from_numpy(xy.reshape(2 * F, n_max, 2)).cuda(),
                  torch.from_numpy(count.reshape(2 * F)).cuda(), f0, f1, seeds)
    kp = res.tracked_kp.cpu().numpy()
    orc.bf_match = lambda a, b, norm=7: c_oracle.match_crosscheck(a, b, norm)

    def frame(i):
        n0, n1 = count[i]
        return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])

    for p in range(F - 1):
        np.random.seed(int(seeds[p]))
        with np.errstate(all="ignore"):
            want = orc.track_pair(frame(p), frame(p + 1))
        T = len(want["kp"])
        assert res.stats[p, 0] == T and np.array_equal(kp[p, :T], want["kp"]), "tracked sets differ"
        assert np.array_equal(res.mask[p, :T], want["mask"]), "supporter masks differ"
        err = np.linalg.norm(res.pose[p] - want["T"]) / np.linalg.norm(want["T"])
        assert err <= 1e-5, f"pose differs: {err}"
    print(f"smoke ok: {F - 1} pairs, tracked {res.stats[:, 0].tolist()}, supporters {res.stats[:, 1].tolist()}")

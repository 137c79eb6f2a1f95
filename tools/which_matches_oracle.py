"""For two builds of libslam_amd: the frame pairs on which their results differ, and which of the two agrees with the
CPU oracle there (tracked count, supporter mask, iterations).  Development tool (imports oracle/ as the checker).

    python tools/which_matches_oracle.py tune/lib_a.so tune/lib_b.so [n_frames] [n_features]
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import compare_libs  # noqa: E402


def main():
    a, b = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 800
    outs = []
    for k, lib in enumerate((a, b)):
        out = os.path.join(ROOT, "gpurun_out", f"which_{k}.npz")
        subprocess.check_call([sys.executable, "-c", compare_libs.WORKER, str(F), str(N), out, "movie"],
                              env={**os.environ, "SLAM_AMD_LIB": lib})
        outs.append(np.load(out))
    diff = [p for p in range(F - 1) if not (np.array_equal(outs[0]["mask"][p], outs[1]["mask"][p])
                                             and np.array_equal(outs[0]["stats"][p], outs[1]["stats"][p]))]
    print("pairs", F - 1, "differing", diff)
    import c_oracle
    import slam_oracle as orc
    from slam_amd.synth import SynthConfig, SynthSequence
    orc.bf_match = lambda x, y, norm=7: c_oracle.match_crosscheck_w64(c_oracle.pad64(x), c_oracle.pad64(y), norm)
    cfg = SynthConfig(n_frames=F, n_features=N, seed=99, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    fr = lambda i: dict(dsc_l=desc[i, 0, :count[i, 0], :61], dsc_r=desc[i, 1, :count[i, 1], :61],
                        xy_l=xy[i, 0, :count[i, 0]], xy_r=xy[i, 1, :count[i, 1]])
    for p in diff[:12]:
        np.random.seed(1000 + p)
        st = {}
        with np.errstate(all="ignore"):
            w = orc.track_pair(fr(p), fr(p + 1), stats=st)
        n = len(w["kp"])
        res = []
        for o in outs:
            res.append((bool(np.array_equal(o["mask"][p][:n], w["mask"])), int(o["stats"][p][2]), int(o["stats"][p][1])))
        print("pair", p, "oracle iterations", st["iterations"], "supporters", int(w["mask"].sum()),
              "| lib A (mask equal, iterations, supporters)", res[0], "| lib B", res[1])


if __name__ == "__main__":
    main()

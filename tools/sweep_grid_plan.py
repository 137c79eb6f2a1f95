"""Matcher time against the grid plan (A rows per lane, B splits) for batches of different sizes: the data behind
plan_grid() in slam_amd/csrc/match.hip.  Forces plans through SLAM_MATCH_PLAN (read by the library at every call).

    python tools/sweep_grid_plan.py > profiles/r2_matcher_grid_plan_sweep.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd.pipeline import ShardTracker, consecutive_pairs  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence, stress_config  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


ROWS = tuple(int(x) for x in os.environ.get('SWEEP_ROWS', '4,2').split(','))
SPLITS = tuple(int(x) for x in os.environ.get('SWEEP_SPLITS', '1,2,3,4,5,6,8,12,16').split(','))
FRAMES = tuple(int(x) for x in os.environ.get('SWEEP_FRAMES', '6,28,55,109,217,433,864,1726,3450').split(','))


def main():
    plans = [(r, ns) for r in ROWS for ns in SPLITS]
    seq2k = SynthSequence(SynthConfig(n_frames=3450, n_features=2000, seed=1234))
    seq8k = SynthSequence(stress_config())
    work = [("2000 features", seq2k, F) for F in FRAMES] + [("8000 features", seq8k, F) for F in (2, 5, 16, 31)]
    print("# matcher stage ms (all-pairs + resolve, HIP events) per plan; default = the library's own choice")
    print("# workload                      problems | default | " + " ".join(f"{r}x{ns:<4d}" for r, ns in plans))
    for name, seq, F in work:
        first = 10 if "8000" in name else 0
        desc, xy, count = seq.frames(first, first + F)
        n_max = seq.cfg.n_max
        args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
        f0, f1 = consecutive_pairs(F)
        trk = ShardTracker(F, F - 1, n_max, n_chunks=1)
        seeds = 1000 + np.arange(F - 1)

        def timed():
            for _ in range(2):
                trk.run(*args, f0, f1, seeds, unpack_mask=False)
            return float(np.median([trk.run(*args, f0, f1, seeds, unpack_mask=False).timing_ms[0] for _ in range(7)]))

        os.environ.pop("SLAM_MATCH_PLAN", None)
        base = timed()
        row = []
        for r, ns in plans:
            os.environ["SLAM_MATCH_PLAN"] = f"{r},{ns}"
            try:
                row.append(timed())
            except Exception:                       # the plan needs more workspace than the tracker sized
                row.append(float("nan"))
        os.environ.pop("SLAM_MATCH_PLAN", None)
        best = int(np.nanargmin(row))
        print(f"{name:14s} {F:5d} frames {2 * F - 1:9d} | {base:7.3f} | " + " ".join(f"{v:6.3f}" for v in row) +
              f" | best {plans[best][0]}x{plans[best][1]} {row[best]:.3f}", flush=True)


if __name__ == "__main__":
    main()

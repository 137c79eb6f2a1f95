"""Latency of the reference-shaped PER-PAIR entry points (what a caller that keeps the reference's own loop over frame
pairs pays per iteration): read_and_rec_match + find_features_in_consecutive_frames (utils/utills.py:386-396, :876-925)
on bench-sized frames, detections from an in-memory source.  Needs a GPU.

    python tools/bench_per_pair_api.py [n_pairs] [n_features]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import utills as U  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    seq = SynthSequence(SynthConfig(n_frames=P + 1, n_features=N, seed=1234))
    desc, xy, count = seq.frames(0, P + 1)
    U.set_detection_source(U.ArrayDetections(desc, xy, count))
    np.random.seed(3)

    def one(i, prev):
        cur = U.read_and_rec_match(i + 1)
        out = U.find_features_in_consecutive_frames(*prev, *cur)
        return cur, out

    prev = U.read_and_rec_match(0)
    prev, _ = one(0, prev)                                   # warm
    t0 = time.perf_counter()
    t_read = t_track = 0.0
    for i in range(1, P):
        a = time.perf_counter()
        cur = U.read_and_rec_match(i + 1)
        b = time.perf_counter()
        out = U.find_features_in_consecutive_frames(*prev, *cur)
        t_read, t_track, prev = t_read + b - a, t_track + time.perf_counter() - b, cur
    dt = (time.perf_counter() - t0) / (P - 1)
    print(f"read_and_rec_match {t_read / (P - 1) * 1e3:.2f} ms, find_features_in_consecutive_frames {t_track / (P - 1) * 1e3:.2f} ms")
    print(f"per pair: {dt * 1e3:.2f} ms  ({1 / dt:.0f} pairs/s), supporters of the last pair {int(np.sum(out[4]))}")
    pr = cProfile.Profile()
    pr.enable()
    prev2 = U.read_and_rec_match(0)
    for i in range(3):
        prev2, _ = one(i, prev2)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()

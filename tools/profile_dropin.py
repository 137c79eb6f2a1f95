"""Where the reference-shaped whole-movie call spends its host time: cProfile of
utills.find_features_in_consecutive_frames_whole_movie() + DataBase(...) (DataDirectory/Data.py:42-43 of the reference)
on the bench movie, detections starting in host memory.  Needs a GPU.

    python tools/profile_dropin.py [n_frames] [n_features]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import utills  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence  # noqa: E402
from slam_amd.types import DataBase  # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 3450
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    seq = SynthSequence(SynthConfig(n_frames=F, n_features=N, seed=1234))
    desc, xy, count = seq.frames(0, F)
    utills.set_detection_source(utills.ArrayDetections(desc, xy, count))

    def call():
        np.random.seed(7)
        return DataBase(*utills.find_features_in_consecutive_frames_whole_movie())

    call()                                                   # warm: library load, allocations, first launches
    t0 = time.perf_counter()
    db = call()
    print(f"whole call {time.perf_counter() - t0:.3f} s, {db.get_num_tracks()} tracks")
    pr = cProfile.Profile()
    pr.enable()
    call()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(30)


if __name__ == "__main__":
    main()

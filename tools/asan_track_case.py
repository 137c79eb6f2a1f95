"""Case files for tools/asan_track_driver (the whole-shard entry point under host AddressSanitizer + UBSan): a small
synthetic sequence, the SlamTrackConfig bytes, and the outputs of the ordinary library on it.

    python tools/asan_track_case.py <out dir>        (GPU box; writes case_*.bin)

Layout of a file: int32 [F, P, N, sizeof(SlamTrackConfig)], the config struct, desc u8 [2F][N][64], xy f32 [2F][N][2],
count i32 [2F], f0 i32 [P], f1 i32 [P], seeds u32 [P], then the expected pose f64 [P][12], stats i32 [P][8],
mask u64 [P][N/64], tracked_kp i32 [P][N][4]."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd.config import HotPathConfig  # noqa: E402
from slam_amd.pipeline import ShardTracker, consecutive_pairs  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence  # noqa: E402


def write_case(path, F, n_features, chunks, cfg, far_pairs=False, seed=5, **synth):
    sc = SynthConfig(n_frames=F + 2, n_features=n_features, seed=seed, spawn_per_frame=max(8, n_features // 10),
                     ragged=min(33, n_features // 3), **synth)
    seq = SynthSequence(sc)
    desc, xy, count = seq.frames(0, F)
    N = sc.n_max
    f0, f1 = consecutive_pairs(F)
    if far_pairs:                                   # a loop-closure-like list: unrelated frames, a frame with itself
        f0 = np.concatenate([f0, [0, 3, 2]]).astype(np.int32)
        f1 = np.concatenate([f1, [F - 1, 3, 0]]).astype(np.int32)
    P = len(f0)
    seeds = (1000 + np.arange(P)).astype(np.uint32)
    trk = ShardTracker(F, P, N, cfg, n_chunks=chunks)
    res = trk.run(torch.from_numpy(desc.reshape(2 * F, N, 64)).cuda(), torch.from_numpy(xy.reshape(2 * F, N, 2)).cuda(),
                  torch.from_numpy(count.reshape(2 * F)).cuda(), f0, f1, seeds, unpack_mask=False)
    with open(path, "wb") as f:
        f.write(np.array([F, P, N, ctypes.sizeof(trk._c)], np.int32).tobytes())
        f.write(bytes(trk._c))
        for a, dt in ((desc, np.uint8), (xy, np.float32), (count, np.int32), (f0, np.int32), (f1, np.int32), (seeds, np.uint32),
                      (res.pose, np.float64), (res.stats, np.int32), (res.mask, np.uint64), (res.tracked_kp.cpu().numpy(), np.int32)):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
    print(path, "F", F, "P", P, "N", N, "tracked", res.stats[:, 0].tolist()[:6], "iterations", res.stats[:, 2].tolist()[:6],
          "draws", res.stats[:, 3].tolist()[:6], "ok", int((res.stats[:, 5] & 1).sum()))


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    write_case(os.path.join(out, "case_a.bin"), 9, 500, 1, HotPathConfig())
    write_case(os.path.join(out, "case_b.bin"), 14, 700, 2, HotPathConfig(), far_pairs=False)
    write_case(os.path.join(out, "case_c.bin"), 8, 300, 1, HotPathConfig(), far_pairs=True)
    knn = HotPathConfig(match_method="REGULAR KNN")
    write_case(os.path.join(out, "case_d.bin"), 7, 400, 3, knn)
    write_case(os.path.join(out, "case_e.bin"), 3, 20, 1, HotPathConfig())       # tiny: few or no tracked points
    # many outliers that survive the stereo-row test: the adaptive loop runs several rounds of several hundred draws
    write_case(os.path.join(out, "case_f.bin"), 7, 900, 2, HotPathConfig(), gross_frac=0.55, gross_sigma=25.0, gross_x_only=True)
    write_case(os.path.join(out, "case_g.bin"), 5, 600, 1, HotPathConfig(fixed_iterations=700), gross_frac=0.3, gross_x_only=True)


if __name__ == "__main__":
    main()

"""Runs the same shard through two builds of libslam_amd (SLAM_AMD_LIB) in separate processes and compares every
output bit for bit: the check that a kernel rewrite meant to keep the arithmetic really did.

    python tools/compare_libs.py tune/lib_old.so slam_amd/lib/libslam_amd.so [n_frames] [n_features] [movie|loops|fixed]

movie: consecutive pairs (one RANSAC round); loops: every frame against near and unrelated earlier frames (capped
adaptive loops, several rounds); fixed: consecutive pairs with a fixed budget of 3000 iterations.
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from slam_amd.pipeline import ShardTracker, consecutive_pairs
from slam_amd.synth import SynthConfig, SynthSequence
F, N, out, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
from slam_amd.config import HotPathConfig
cfg = SynthConfig(n_frames=F, n_features=N, seed=99, ragged=True)
desc, xy, count = SynthSequence(cfg).frames(0, F)
n_max = cfg.n_max
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
f0, f1 = consecutive_pairs(F)
hp = HotPathConfig(fixed_iterations=3000) if mode == "fixed" else HotPathConfig()
if mode == "loops":
    rng = np.random.default_rng(1)
    cur = np.repeat(np.arange(40, F), 4)
    cand = np.stack([np.arange(40, F) - 2, np.arange(40, F) - 6, rng.integers(0, 20, F - 40), rng.integers(0, 20, F - 40)], 1).reshape(-1)
    f0, f1 = cand.astype(np.int32), cur.astype(np.int32)
res = ShardTracker(F, len(f0), n_max, hp, n_chunks=2).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                                         dev(count.reshape(2 * F)), f0, f1, 1000 + np.arange(len(f0)))
np.savez(out, pose=res.pose, stats=res.stats[:, :6], mask=res.mask, kp=res.tracked_kp.cpu().numpy())
''' % ROOT


def main():
    a, b = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
    F = sys.argv[3] if len(sys.argv) > 3 else "300"
    N = sys.argv[4] if len(sys.argv) > 4 else "1500"
    mode = sys.argv[5] if len(sys.argv) > 5 else "movie"
    outs = []
    for k, lib in enumerate((a, b)):
        out = os.path.join(ROOT, "gpurun_out", f"compare_{k}.npz")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call([sys.executable, "-c", WORKER, F, N, out, mode], env={**os.environ, "SLAM_AMD_LIB": lib})
        outs.append(np.load(out))
    same = {k: bool(np.array_equal(outs[0][k], outs[1][k], equal_nan=True)) for k in outs[0].files}
    worst = float(np.nanmax(np.abs(outs[0]["pose"] - outs[1]["pose"])))
    st = outs[1]["stats"]
    print("mode", mode, "iterations min/mean/max", int(st[:, 2].min()), float(st[:, 2].mean()), int(st[:, 2].max()),
          "draws max", int(st[:, 3].max()))
    print("pairs", len(outs[0]["pose"]), "bit-identical:", same, "max |pose difference|", worst)
    sys.exit(0 if all(same.values()) else 1)


if __name__ == "__main__":
    main()

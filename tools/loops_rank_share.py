"""One rank's share of BASELINE configs[4] (loop-closure candidate validation, 3450 frames x 5 candidates) at world
sizes 2 / 4 / 8, timed on ONE card for both plans of slam_amd.dist.track_pairs_sharded:

  one  the rank stereo-matches every frame its pair block touches (no data-path collective)
  two  the rank stereo-matches only its block of the frame list; the other frames' keys come from the all-gather
       (here: computed before the timed loop, standing in for the other ranks' contributions -- the exchange itself,
       28 MB over xGMI, is not in the figure)

What the busiest rank's step takes bounds the N-GPU figure: projected pairs/s = all 17 220 pairs / that time.  No
multi-GPU hardware is needed (or used); results of the two plans are compared bit for bit.

    python tools/loops_rank_share.py [--frames 3450] [--features 2000] [--steps 10] [--worlds 2,4,8]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import dist as sdist  # noqa: E402
from slam_amd import pipeline  # noqa: E402
from slam_amd.config import HotPathConfig  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence, loop_candidate_pairs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3450)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--worlds", default="2,4,8")
    args = ap.parse_args()
    F = args.frames
    scfg = SynthConfig(n_frames=F, n_features=args.features, seed=1234)
    n_max = scfg.n_max
    seq = SynthSequence(scfg)
    desc, xy, count = seq.frames(0, F)
    d_desc = torch.from_numpy(desc.reshape(2 * F, n_max, 64)).cuda()
    d_xy = torch.from_numpy(xy.reshape(2 * F, n_max, 2)).cuda()
    d_count = torch.from_numpy(count.reshape(2 * F)).cuda()
    hp = HotPathConfig()
    g0, g1 = loop_candidate_pairs(F)
    P_total = len(g0)
    table = pipeline.stereo_keys(d_desc, d_count, hp)                       # every frame's keys (the all-gather's result)

    def sub(frames):
        idx = torch.from_numpy(np.stack([2 * frames, 2 * frames + 1], 1).reshape(-1)).cuda()
        return d_desc[idx].contiguous(), d_xy[idx].contiguous(), d_count[idx].contiguous()

    print(f"{F} frames x {args.features} features, {P_total} candidate pairs; per-rank step on one card, busiest rank of each world")
    print("world plan  rank  frames held  stereo + temporal problems   ms/step   projected pairs/s at that world   vs one GPU")
    one_gpu_ms = None
    for world in [1] + [int(w) for w in args.worlds.split(",")]:
        costs = []
        for r in range(world):
            a, b = sdist.shard_list(P_total, r, world)
            costs.append(len(np.unique(np.concatenate([g0[a:b], g1[a:b]]))) + (b - a))
        rank = int(np.argmax(costs))
        a, b = sdist.shard_list(P_total, rank, world)
        fa, fb = sdist.block_range(F, rank, world)
        seeds = 9000 + a + np.arange(b - a)
        ref = None
        for plan in (("one",) if world == 1 else ("one", "two")):
            used, f0, f1 = sdist.frames_of_block(g0, g1, a, b)
            if plan == "two":
                used = np.union1d(used, np.arange(fa, fb))
                f0, f1 = np.searchsorted(used, g0[a:b]).astype(np.int32), np.searchsorted(used, g1[a:b]).astype(np.int32)
            dd, dx, dc = sub(used)
            Fl = len(used)
            trk = pipeline.ShardTracker(Fl, b - a, n_max, hp, n_chunks=1)
            i0 = int(np.searchsorted(used, fa))
            used_dev = torch.from_numpy(used.astype(np.int64)).cuda()
            foreign = table.clone()
            foreign[fa:fb] = -1                                               # the rank's own block is computed in the step

            def step():
                keys = None
                if plan == "two":
                    own = pipeline.stereo_keys(dd[2 * i0:2 * (i0 + fb - fa)], dc[2 * i0:2 * (i0 + fb - fa)], hp)
                    foreign[fa:fb] = own                                      # where the all-gather would put it
                    keys = foreign[used_dev].contiguous()
                return trk.run(dd, dx, dc, f0, f1, seeds, unpack_mask=False, stereo_key=keys)

            for _ in range(args.warmup):
                res = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                res = step()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / args.steps
            if world == 1:
                one_gpu_ms = ms
            if ref is None:
                ref = res
            else:
                same = (np.array_equal(ref.pose, res.pose, equal_nan=True) and np.array_equal(ref.stats, res.stats)
                        and np.array_equal(ref.mask, res.mask))
                assert same, "the two plans disagree"
            n_st = (fb - fa) if plan == "two" else Fl
            print(f"{world:5d} {plan:4s} {rank:5d} {Fl:12d} {n_st:8d} + {b - a:6d} {ms:19.3f} {P_total / ms * 1e3:22.0f} "
                  f"{one_gpu_ms / ms:20.2f}x" + ("   (results equal plan one's bit for bit)" if plan == "two" else ""))
            del trk, dd, dx, dc


if __name__ == "__main__":
    main()

"""Per-call latency of reference-shaped loop-closure validation calls (find_loop_cand_by_consensus, utils/utills.py:1118-1146,
called with at most 3 candidates from PoseGraph.py:403; BASELINE configs[4] asks for 5): one current frame against C
candidate frames = C + 1 stereo problems + C temporal problems in one slam_track_pairs call, detections resident.

    python tools/bench_small_batches.py [label]

Prints one JSON line: for C in 1, 3, 5 the ms per call, the matcher's share, and a digest of the results (keys, masks,
poses) so that two library versions can be compared for identical output."""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd.pipeline import ShardTracker  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def main(label):
    cfg = SynthConfig(n_frames=120, n_features=2000, seed=1234)
    seq = SynthSequence(cfg)
    n_max = cfg.n_max
    out = {"label": label, "workload": "current frame 100 vs candidates (98, 95, 40, 10, 70)[:C], 2000 features/image"}
    for C in (1, 3, 5):
        frames = [100] + [98, 95, 40, 10, 70][:C]
        per = [seq.frame(f) for f in frames]
        desc = np.stack([p["desc"] for p in per]); xy = np.stack([p["xy"] for p in per]); count = np.stack([p["count"] for p in per])
        F = len(frames)
        args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
        f0 = np.arange(1, F, dtype=np.int32)            # candidate = "frame 0" of the pair
        f1 = np.zeros(C, np.int32)                      # current frame = "frame 1"
        trk = ShardTracker(F, C, n_max)
        seeds = 77 + np.arange(C)
        for _ in range(5):
            res = trk.run(*args, f0, f1, seeds, unpack_mask=False)
        torch.cuda.synchronize()
        n = 200
        tm = []
        t0 = time.perf_counter()
        for _ in range(n):
            res = trk.run(*args, f0, f1, seeds, unpack_mask=False)
            tm.append(res.timing_ms.copy())
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        tm = np.mean(tm, axis=0)
        h = hashlib.sha256()
        for a in (res.stats, res.mask, res.tracked_kp.cpu().numpy()[:, :int(res.stats[:, 0].max())]):
            h.update(np.ascontiguousarray(a).tobytes())
        out[f"C{C}"] = {"ms_per_call": round(ms, 4), "match_ms": round(float(tm[0]), 4), "ransac_rounds_ms": round(float(tm[3]), 4),
                        "tracked": res.stats[:, 0].tolist(), "supporters": res.stats[:, 1].tolist(), "digest": h.hexdigest()[:16]}
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "")

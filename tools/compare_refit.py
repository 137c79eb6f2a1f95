"""Development check of a refit kernel change: bench shard through the pipeline, digest + timing of the refit stage."""
import hashlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd.pipeline import ShardTracker, consecutive_pairs
from slam_amd.synth import SynthConfig, SynthSequence
F = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = SynthConfig(n_frames=F, n_features=2000, seed=1234)
desc, xy, count = SynthSequence(cfg).frames(0, F)
n_max = cfg.n_max
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
f0, f1 = consecutive_pairs(F)
trk = ShardTracker(F, F - 1, n_max, n_chunks=1)
for _ in range(3):
    res = trk.run(*args, f0, f1, 1000 + np.arange(F - 1))
tm = np.mean([trk.run(*args, f0, f1, 1000 + np.arange(F - 1)).timing_ms for _ in range(10)], axis=0)
np.save("gpurun_out/refit_pose_%s.npy" % (sys.argv[2] if len(sys.argv) > 2 else "x"), res.pose)
print(json.dumps({"mask_refit_ms": round(float(tm[4]), 4), "call_ms": round(float(tm[5]), 3),
                  "pose_digest": hashlib.sha256(res.pose.tobytes()).hexdigest()[:12], "supporters_max": int(res.stats[:, 1].max())}))

"""Trial steps per pair of the Levenberg-Marquardt refit on bench-shaped pairs, and how the refit's time moves with the
stopping threshold.  Needs a tuning build whose refit status byte carries the count:

    bash tools/build_variant.sh probe -DSLAM_REFIT_PROBE=1
    SLAM_AMD_LIB=tune/libslam_amd_probe.so python tools/refit_evals_hist.py

The pipeline's intermediate arrays are rebuilt through the per-stage entry points: tracked key-points -> coordinates ->
triangulation -> the winning draw's AP3P pose (slam_randint_draws + slam_p3p_hypotheses at stats' best draw) -> refit."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import ops, pipeline  # noqa: E402
from slam_amd.config import HotPathConfig  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 800
scfg = SynthConfig(n_frames=F, n_features=2000, seed=1234)
desc, xy, count = SynthSequence(scfg).frames(0, F)
n_max = scfg.n_max
f0, f1 = pipeline.consecutive_pairs(F)
hp = HotPathConfig()
P = F - 1
seeds = 1000 + np.arange(P)
trk = pipeline.ShardTracker(F, P, n_max, hp)
d_desc = torch.from_numpy(desc.reshape(2 * F, n_max, 64)).cuda()
d_xy = torch.from_numpy(xy.reshape(2 * F, n_max, 2)).cuda()
d_count = torch.from_numpy(count.reshape(2 * F)).cuda()
res = trk.run(d_desc, d_xy, d_count, f0, f1, seeds, unpack_mask=False)
T = torch.from_numpy(res.stats[:, 0].copy()).cuda()
kp = res.tracked_kp.long()                                                  # [P, n_max, 4]: L0, R0, L1, R1
pf0, pf1 = torch.from_numpy(f0).long().cuda(), torch.from_numpy(f1).long().cuda()
xyf = d_xy.reshape(F, 2, n_max, 2)


def gather(frame, side, col):
    idx = kp[:, :, col].clamp(0, n_max - 1)
    return torch.gather(xyf[frame, side], 1, idx[:, :, None].expand(-1, -1, 2))


xy0 = torch.cat([gather(pf0, 0, 0), gather(pf0, 1, 1)], 2).contiguous()
xy1 = torch.cat([gather(pf1, 0, 2), gather(pf1, 1, 3)], 2).contiguous()
Pm, Qm = hp.K @ hp.M1, hp.K @ hp.M2
X = ops.triangulate(Pm, Qm, xy0, T)
sample = ops.randint_draws(torch.from_numpy(seeds.astype(np.int32)).cuda(), T, torch.zeros(P, dtype=torch.int32, device="cuda"), 64)
pose, status = ops.p3p_hypotheses(hp.K, X, xy1, T, sample)
best = torch.from_numpy(res.stats[:, 4].copy()).long().cuda().clamp(0, 63)
pose0 = pose[torch.arange(P, device="cuda"), best].contiguous()
words = torch.from_numpy(res.mask.copy()).cuda()
out, st = ops.refit_pnp(hp.K, X, xy1, T, words, pose0)
torch.cuda.synchronize()
ok = (res.stats[:, 5] & 1) != 0
err = np.linalg.norm((out.cpu().numpy() - res.pose)[ok].reshape(-1, 12), axis=1) / np.linalg.norm(res.pose[ok].reshape(-1, 12), axis=1)
print(f"{P} pairs; rebuilt refit against the pipeline's poses: worst relative deviation {err.max():.3g}")
ev = st.cpu().numpy()[ok]
print("trial steps per pair (each = one Cholesky solve + one cost pass; accepted ones are preceded by a Jacobian pass):")
for k in range(1, int(ev.max()) + 1):
    print(f"  {k:2d}: {(ev == k).mean():.4f}")
print("mean", ev.mean())
t0 = time.perf_counter()
for _ in range(20):
    ops.refit_pnp(hp.K, X, xy1, T, words, pose0)
torch.cuda.synchronize()
print(f"refit of {P} pairs: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
ref = os.environ.get("SLAM_REFIT_REF")                      # poses of another build, saved by an earlier run
if ref and os.path.exists(ref):
    want = np.load(ref)
    dev = np.linalg.norm((out.cpu().numpy() - want)[ok].reshape(-1, 12), axis=1) / np.linalg.norm(want[ok].reshape(-1, 12), axis=1)
    print(f"against {ref}: relative deviation median {np.median(dev):.3g}, worst {dev.max():.3g}")
elif ref:
    np.save(ref, out.cpu().numpy())

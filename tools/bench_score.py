"""Consensus scoring alone: slam_score_hypotheses (FP64) against the packed single-precision path, on scenes shaped
like the bench configurations' RANSAC rounds.  python tools/bench_score.py  (on the GPU box)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import slam_oracle as orc
from slam_amd import ops

T0 = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.004, -0.02, 0.003])), np.array([[0.03], [-0.01], [-0.9]])])

def case(S, n, H, garbage, seed=1):
    rng = np.random.default_rng(seed)
    stride = (n + 63) // 64 * 64
    X = np.stack([rng.uniform(-30, 30, (S, n)), rng.uniform(-5, 4, (S, n)), rng.uniform(4, 100, (S, n))], 2)
    g = rng.random((S, n)) < garbage                       # mismatched correspondences triangulate anywhere
    X[g] = rng.normal(0, 1, (int(g.sum()), 3)) * 10.0 ** rng.uniform(-1, 4, (int(g.sum()), 1))
    Xs = np.zeros((S, stride, 3)); xy1 = np.zeros((S, stride, 4), np.float32)
    for s in range(S):
        l1 = orc.project(X[s], orc.K @ T0) + rng.normal(0, 0.5, (n, 2))
        r1 = orc.project(X[s], orc.K @ orc.compose(T0, orc.M2)) + rng.normal(0, 0.5, (n, 2))
        Xs[s, :n] = X[s]; xy1[s, :n] = np.hstack([l1, r1])
    xy1[~np.isfinite(xy1)] = 0
    Ts = np.zeros((S, H, 3, 4))
    rv = np.array([0.004, -0.02, 0.003]) + rng.normal(0, 1, (S, H, 3)) * np.where(rng.random((S, H, 1)) < 0.3, 0.003, 0.5)
    for s in range(S):
        for h in range(H):
            Ts[s, h] = np.hstack([orc.rodrigues_vec_to_mat(rv[s, h]), (T0[:, 3] + rng.normal(0, 0.5, 3))[:, None]])
    return [torch.from_numpy(a).cuda() for a in (Xs, xy1, np.full(S, n, np.int32), Ts, np.ones((S, H), np.uint8))]

def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps

for name, S, n, H, garbage in (("round 0 (64 draws)", 3449, 1196, 64, 0.02), ("loops later round", 1024, 569, 1024, 0.3),
                               ("stress", 30, 5973, 10048, 0.6), ("stress clean", 30, 5973, 10048, 0.0)):
    if len(sys.argv) > 1 and sys.argv[1] not in name: continue
    Xs, xy1, cnt, Ts, st = case(S, n, H, garbage)
    a = timeit(lambda: ops.score_hypotheses(orc.K, orc.M2, Xs, xy1, cnt, Ts, st, 4.0))
    packed, scale = ops.score_pack_points(Xs, xy1, cnt)
    b = timeit(lambda: ops.score_hypotheses_packed(orc.K, orc.M2, Xs, xy1, packed, scale, cnt, Ts, st, 4.0))
    r, _ = ops.score_hypotheses(orc.K, orc.M2, Xs, xy1, cnt, Ts, st, 4.0)
    p = ops.score_hypotheses_packed(orc.K, orc.M2, Xs, xy1, packed, scale, cnt, Ts, st, 4.0)
    tests = Xs.shape[0] * n * H
    print(f"{name:20s} segs {Xs.shape[0]:5d} n {n:5d} H {H:6d}  fp64 {a:8.3f} ms ({tests / a / 1e9:6.2f} T tests/s)  packed {b:8.3f} ms ({tests / b / 1e9:6.2f} T tests/s)  equal {bool((r == p).all())}  mean count {float(r.float().mean()):.1f}", flush=True)

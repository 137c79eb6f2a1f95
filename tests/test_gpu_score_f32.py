"""Consensus counts through the single-precision pre-classification (slam_score_pack_points +
slam_score_hypotheses_packed, csrc/score.hip) against the CPU oracle's consensus_match (utils/utills.py:533-545,
:659-736) and against slam_score_hypotheses: identical, bit for bit -- on ordinary data, on points built to sit inside
the single-precision band around the 4-pixel radius (so that the FP64 settling paths and their overflow paths run), and on
everything irregular (non-finite and denormal-range values, the camera plane, failed and NaN hypotheses, empty segments).
All three kernels: wavefront per 4 / 2 hypotheses (round 0 of the RANSAC loop) and lane per hypothesis (later rounds)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import slam_oracle as orc
from slam_amd import ops

pytestmark = pytest.mark.gpu

T0 = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.004, -0.02, 0.003])), np.array([[0.03], [-0.01], [-0.9]])])


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def both_paths(K, M2, Xs, xy1, count, Ts, status, thr=4.0):
    """(counts of slam_score_hypotheses, counts of the packed path) for float32 observations"""
    d = [dev(Xs), dev(xy1.astype(np.float32)), dev(np.asarray(count, np.int32)), dev(Ts), dev(status)]
    ref, _ = ops.score_hypotheses(K, M2, *d, thr)
    packed, scale = ops.score_pack_points(d[0], d[1], d[2])
    got = ops.score_hypotheses_packed(K, M2, d[0], d[1], packed, scale, d[2], d[3], d[4], thr)
    torch.cuda.synchronize()
    return ref.cpu().numpy(), got.cpu().numpy()


def oracle_counts(K, M2, X, l1, r1, Ts, status=None):
    out = np.zeros(len(Ts), np.int64)
    with np.errstate(all="ignore"):
        for h, T in enumerate(Ts):
            if status is None or status[h]:
                out[h] = orc.consensus_match(K, T, M2, X, l1.astype(np.float64), r1.astype(np.float64), 2).sum()
    return out


def scene(rng, n, noise=2.5):
    X = np.stack([rng.uniform(-30, 30, n), rng.uniform(-5, 4, n), rng.uniform(-5, 100, n)], 1)
    l1 = (orc.project(X, orc.K @ T0) + rng.normal(0, noise, (n, 2))).astype(np.float32)
    r1 = (orc.project(X, orc.K @ orc.compose(T0, orc.M2)) + rng.normal(0, noise, (n, 2))).astype(np.float32)
    return X, l1, r1


def poses(rng, H):
    Ts = np.zeros((H, 3, 4))
    for h in range(H):
        sr, st = ((0.002, 0.02), (0.02, 0.3), (0.8, 5.0))[h % 3]        # near the truth / partly off / garbage
        Ts[h] = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.004, -0.02, 0.003]) + rng.normal(0, sr, 3)),
                           (T0[:, 3] + rng.normal(0, st, 3))[:, None]])
    return Ts


@pytest.mark.parametrize("H", [5, 8, 64, 96, 300, 1000])
def test_packed_counts_equal_the_oracle_on_ordinary_data(H):
    """5 / 8 hypotheses: two per wavefront; 64 / 96: four per wavefront (RANSAC round 0); 300 / 1000: one per lane."""
    rng = np.random.default_rng(100 + H)
    sizes = [0, 1, 63, 64, 65, 700, 1900]
    S, stride = len(sizes), 1920
    Xs, xy1 = np.zeros((S, stride, 3)), np.zeros((S, stride, 4), np.float32)
    Ts = np.zeros((S, H, 3, 4))
    status = (rng.random((S, H)) > 0.1).astype(np.uint8)
    want = np.zeros((S, H), np.int64)
    for s, n in enumerate(sizes):
        X, l1, r1 = scene(rng, n)
        Ts[s] = poses(rng, H)
        Ts[s, 1, 0, 1], Ts[s, 2, 1, 3], Ts[s, 3, 2, 0], Ts[s, 4, 0, 3] = np.nan, np.inf, -np.inf, np.nan
        Ts[s, 0] = np.nan if s % 2 else Ts[s, 0]                          # what a degenerate sample produces
        status[s, :5] = 1
        if n > 7:
            X[7] = -np.linalg.solve(Ts[s, 4 % H, :, :3], T0[:, 3]) + np.array([0.3, 0.1, 0.0])
        Xs[s, :n], xy1[s, :n] = X, np.hstack([l1, r1])
        want[s] = oracle_counts(orc.K, orc.M2, X, l1, r1, Ts[s], status[s]) if n else 0
    ref, got = both_paths(orc.K, orc.M2, Xs, xy1, sizes, Ts, status)
    assert np.array_equal(ref, want)
    assert np.array_equal(got, want)
    assert want[-1].max() > 100


@pytest.mark.parametrize("H", [4, 64, 300])
def test_points_inside_the_single_precision_band_are_settled_in_fp64(H):
    """Every point of the segment lies within a few 1e-5 px of the radius for hypothesis 0 (the observations are the
    hypothesis' own projections pushed out by 4 px, rounded to float32: +-3e-5 px of rounding decides which side), so every
    test of that hypothesis is undecided in single precision: the LDS queue of the wavefront kernels overflows (more
    than 64 per wavefront: their FP64 recount runs) and the lane kernel's per-wavefront queue is drained again and again.
    Hypotheses 1, 2: the same with a sprinkling of such points (queues that do not overflow)."""
    rng = np.random.default_rng(7 + H)
    n, stride = 900, 960
    X, _, _ = scene(rng, n)
    X[:, 2] = rng.uniform(4, 80, n)
    Ts = poses(rng, H)
    Ts[0] = T0
    status = np.ones(H, np.uint8)

    def on_the_radius(T, idx):
        ang = rng.uniform(0, 2 * np.pi, len(idx))
        off = 4.0 * np.stack([np.cos(ang), np.sin(ang)], 1)
        pl = orc.project(X[idx], orc.K @ T)
        pr = orc.project(X[idx], orc.K @ orc.compose(T, orc.M2))
        return (pl + off).astype(np.float32), (pr + off * rng.uniform(0, 1, (len(idx), 1))).astype(np.float32)

    l1, r1 = on_the_radius(Ts[0], np.arange(n))
    if H > 2:
        for h, k in ((1, 3), (2, 40)):
            idx = rng.choice(n, k, replace=False)
            # these points sit on hypothesis h's radius instead (hypothesis 0 then sees them far away)
            l1[idx], r1[idx] = on_the_radius(Ts[h], idx)
    want = oracle_counts(orc.K, orc.M2, X, l1, r1, Ts)
    Xs, xy1 = np.zeros((1, stride, 3)), np.zeros((1, stride, 4), np.float32)
    Xs[0, :n], xy1[0, :n] = X, np.hstack([l1, r1])
    ref, got = both_paths(orc.K, orc.M2, Xs, xy1, [n], Ts[None], status[None])
    assert np.array_equal(ref[0], want)
    assert np.array_equal(got[0], want)
    assert 0.2 * n < want[0] < 0.8 * n            # the radius really splits them


@pytest.mark.parametrize("H", [40, 320])
def test_packed_counts_on_irregular_values(H):
    """The unit rectified rig (hx = x, hy = y, hz = z in both cameras): points exactly on the radius and ulps either side,
    on the camera plane, depths whose square under- or overflows in double and in single precision, values outside the
    float32 range and in its denormal range, non-finite points, observations and poses; radius 4, and radii the
    single-precision path does not take (0, negative, 1e-4, 1e7)."""
    K, M2 = np.eye(3), np.hstack([np.eye(3), np.zeros((3, 1))])
    T = np.hstack([np.eye(3), np.zeros((3, 1))])
    u, v = 10.0, 5.0
    pts = []

    def add(x, y, z, ul=u, vl=v, ur=None, vr=v):
        pts.append(((x, y, z), (ul, vl), ((x / z if z != 0.0 and np.isfinite(z) and np.isfinite(x) else 0.0) if ur is None else ur, vr)))

    for d in (0.0, 2.0 ** -50, -2.0 ** -50, 2.0 ** -44, -2.0 ** -44, 1e-8, -1e-8, 1e-6, -1e-6, 1e-5, -1e-5, 1e-3, -1e-3, 0.1, -0.1):
        add(28.0 + d, 10.0, 2.0)                       # left distance 4 + d / 2
        add(20.0 + d, 10.0, 2.0, ul=6.0, ur=14.0)      # right distance 4 + d / 2, left 4 - d / 2: one side decides
    add(20.0, 10.0, 2.0)
    add(3.0, 1.0, 0.0)
    add(0.0, 0.0, 0.0)
    for e in (1e-170, 1e-140, 1e-60, 1e-42, 1e-36, 1e-25, 1e-19, 1e-12, 1e12, 1e19, 1e25, 1e36, 1e39, 1e60, 1e200):
        add(12 * e, 5 * e, e)                          # projects to (12, 5): inside by 2 px, at every magnitude
        add(12 * e, 5 * e, e, ur=17.0)                 # right camera 5 px off
        add(14.1 * e, 5 * e, e)                        # outside by 0.1 px
    add(-24.0, -10.0, -2.0)
    add(np.inf, 1.0, 2.0)
    add(1.0, np.nan, 2.0)
    add(24.0, 10.0, np.inf)
    add(24.0, 10.0, 2.0, ul=np.nan)
    add(24.0, 10.0, 2.0, ur=np.inf)
    add(24.0, 10.0, 2.0, vr=np.nan)
    add(24.0, 10.0, 2.0, ul=3e38, ur=-3e38)
    add(24e-40, 10e-40, 2e-40)                         # float32 denormals
    n, stride = len(pts), 128
    X = np.array([p[0] for p in pts])
    l1 = np.array([p[1] for p in pts]).astype(np.float32)
    r1 = np.array([p[2] for p in pts]).astype(np.float32)
    Xs, xy1 = np.zeros((1, stride, 3)), np.zeros((1, stride, 4), np.float32)
    Xs[0, :n], xy1[0, :n] = X, np.hstack([l1, r1])
    Ts = np.repeat(T[None, None], H, 1)
    Ts[0, 1::2, 0, 3] = 2.0 ** -45
    Ts[0, 2, 0, 0] = 1e-35                             # a matrix entry in the float32 denormal range
    Ts[0, 3, 1, 3] = 1e39                              # ... beyond the float32 range
    Ts[0, 4, 2, 2] = np.nan
    Ts[0, 5, 0, 3] = np.inf
    status = np.ones((1, H), np.uint8)
    status[0, 6] = 0
    for acc, thr in ((2, 4.0), (0, 0.0), (None, -1.0), (0.01, 1e-4), (np.sqrt(1e7), 1e7)):
        want = np.zeros(H, np.int64)
        if acc is not None:
            with np.errstate(all="ignore"):
                for h in range(H):
                    if status[0, h]:
                        P = K @ Ts[0, h]
                        inl = np.ones(n, bool)
                        for obs in (l1, r1):           # both cameras see the same matrix (zero baseline)
                            h3 = X @ P[:, :3].T + P[:, 3]
                            uv = h3[:, :2] / h3[:, 2:]
                            inl &= np.linalg.norm(uv - obs.astype(np.float64), axis=1) <= thr
                        want[h] = inl.sum()
        ref, got = both_paths(K, M2, Xs, xy1, [n], Ts, status, thr)
        assert np.array_equal(ref[0], want), thr
        assert np.array_equal(got[0], want), thr
    assert want.max() > 0


def test_packed_path_on_a_general_rig_takes_the_fp64_kernels():
    Kg = np.array([[700.0, 1.5, 600.0], [0.0, 710.0, 180.0], [0.0, 0.0, 1.0]])
    M2g = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.002, -0.004, 0.001])), np.array([[-0.54], [0.003], [0.002]])])
    rng = np.random.default_rng(5)
    n, H, stride = 500, 70, 512
    X = np.stack([rng.uniform(-30, 30, n), rng.uniform(-5, 4, n), rng.uniform(4, 100, n)], 1)
    l1 = (orc.project(X, Kg @ T0) + rng.normal(0, 2.5, (n, 2))).astype(np.float32)
    r1 = (orc.project(X, Kg @ orc.compose(T0, M2g)) + rng.normal(0, 2.5, (n, 2))).astype(np.float32)
    Ts = poses(rng, H)
    want = oracle_counts(Kg, M2g, X, l1, r1, Ts)
    Xs, xy1 = np.zeros((1, stride, 3)), np.zeros((1, stride, 4), np.float32)
    Xs[0, :n], xy1[0, :n] = X, np.hstack([l1, r1])
    ref, got = both_paths(Kg, M2g, Xs, xy1, [n], Ts[None], np.ones((1, H), np.uint8))
    assert np.array_equal(ref[0], want) and np.array_equal(got[0], want)


@pytest.mark.parametrize("H", [6, 300])
def test_single_precision_overflow_and_cancellation_stay_undecided(H):
    """KITTI rig: points far enough out that fx * x overflows float32 while x itself does not (hx = inf in single
    precision, q = inf: must not read as "outside" when the exact distance is inside), and points whose homogeneous
    coordinates cancel (a camera rotated so that hz is a small difference of large terms).  Counts against the oracle."""
    rng = np.random.default_rng(31 + H)
    pts = []
    for e in (1.0, 1e20, 1e30, 1e34, 1e35, 3e35, 1e36, 1e37, 2.5e38):
        for k in range(6):
            d = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.1, 0.1), 1.0]) * rng.uniform(5, 50)
            pts.append(d * e)
    X = np.array(pts)
    n, stride = len(X), 64
    pl = orc.project(X, orc.K @ T0)
    pr = orc.project(X, orc.K @ orc.compose(T0, orc.M2))
    off = rng.uniform(-5, 5, (n, 2))                       # about half inside the 4 px radius
    l1, r1 = (pl + off).astype(np.float32), (pr + off).astype(np.float32)
    Ts = poses(rng, H)
    Ts[0] = T0
    # a camera looking sideways: the depth of a point straight ahead is a small difference of large terms
    Ts[1] = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.0, np.pi / 2 - 1e-7, 0.0])), np.array([[0.0], [0.0], [1e-3]])])
    want = oracle_counts(orc.K, orc.M2, X, l1, r1, Ts)
    Xs, xy1 = np.zeros((1, stride, 3)), np.zeros((1, stride, 4), np.float32)
    Xs[0, :n], xy1[0, :n] = X, np.hstack([l1, r1])
    ref, got = both_paths(orc.K, orc.M2, Xs, xy1, [n], Ts[None], np.ones((1, H), np.uint8))
    assert np.array_equal(ref[0], want)
    assert np.array_equal(got[0], want)
    assert 10 < want[0] < n - 5

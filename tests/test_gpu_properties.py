"""The same implementation-independent properties as tests/test_oracle_properties.py, asked of the HIP kernels directly
(not through the oracle): AP3P returns the planted pose on exact data, the refit returns a stationary point of the left
reprojection error and the planted pose on exact data, triangulation inverts projection."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from slam_amd import ops
from slam_amd.synth import K_KITTI as K, M1_KITTI as M1, M2_KITTI as M2

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _scene(rng, n, noise=0.0):
    T = np.hstack([Rotation.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix(), rng.normal(0, 0.5, (3, 1)) + [[0.0], [0.0], [-0.8]]])
    X = np.stack([rng.uniform(-15, 15, n), rng.uniform(-4, 3, n), rng.uniform(5, 60, n)], 1)
    h = X @ (K @ T)[:, :3].T + (K @ T)[:, 3]
    return T, X, h[:, :2] / h[:, 2:] + rng.normal(0, noise, (n, 2))


@pytest.mark.parametrize("seed", range(3))
def test_p3p_kernel_returns_the_planted_pose_on_exact_data(seed):
    rng = np.random.default_rng(seed)
    T, X, uv = _scene(rng, 400)
    n, stride, H = len(X), 448, 256
    sample = np.array([rng.choice(n, 4, replace=False) for _ in range(H)], np.int32)[None]
    Xs = np.zeros((1, stride, 3)); Xs[0, :n] = X
    xy1 = np.zeros((1, stride, 4)); xy1[0, :n, :2] = uv                     # float64 observations: the data stay exact
    pose, status = ops.p3p_hypotheses(K, dev(Xs), dev(xy1), dev(np.array([n], np.int32)), dev(sample))
    pose, status = pose.cpu().numpy()[0], status.cpu().numpy()[0]
    ok = status == 1
    err = np.linalg.norm((pose[ok] - T).reshape(-1, 12), axis=1) / np.linalg.norm(T)
    assert ok.sum() >= 0.9 * H and (err < 1e-6).all()
    R = pose[ok][:, :, :3]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-9 and np.abs(np.linalg.det(R) - 1).max() < 1e-9


@pytest.mark.parametrize("seed", range(3))
def test_refit_kernel_returns_a_stationary_point_and_the_planted_pose(seed):
    rng = np.random.default_rng(50 + seed)
    n, stride = 500, 512
    T, X, uv_exact = _scene(rng, n)
    uv = uv_exact + rng.normal(0, 0.5, (n, 2))
    start = np.hstack([Rotation.from_rotvec(rng.normal(0, 2e-3, 3)).as_matrix() @ T[:, :3], T[:, 3:] + rng.normal(0, 0.02, (3, 1))])
    Xs = np.zeros((2, stride, 3)); Xs[:, :n] = X
    xy1 = np.zeros((2, stride, 4)); xy1[0, :n, :2] = uv; xy1[1, :n, :2] = uv_exact
    sel = rng.random(n) < 0.8
    words = ops.bool_to_mask_words(np.stack([sel, sel]), stride)
    pose, status = ops.refit_pnp(K, dev(Xs), dev(xy1), dev(np.array([n, n], np.int32)), dev(words), dev(np.stack([start, start])))
    pose, status = pose.cpu().numpy(), status.cpu().numpy()
    assert list(status) == [1, 1]
    assert np.linalg.norm(pose[1] - T) / np.linalg.norm(T) < 1e-9           # exact data: the planted pose

    def cost(d):
        R = Rotation.from_rotvec(d[:3]).as_matrix() @ pose[0][:, :3]
        P = K @ np.hstack([R, (pose[0][:, 3] + d[3:]).reshape(3, 1)])
        h = X[sel] @ P[:, :3].T + P[:, 3]
        return float(np.sum((h[:, :2] / h[:, 2:] - uv[sel]) ** 2))

    c0 = cost(np.zeros(6))
    for a in range(6):
        e = np.zeros(6); e[a] = 1e-5
        g = (cost(e) - cost(-e)) / 2e-5
        curv = (cost(e) - 2 * c0 + cost(-e)) / 1e-10
        assert curv > 0 and abs(g / curv) < 1e-7, (a, g, curv)               # a Newton step would not move the pose
    hs = X[sel] @ (K @ start)[:, :3].T + (K @ start)[:, 3]
    assert c0 < float(np.sum((hs[:, :2] / hs[:, 2:] - uv[sel]) ** 2))        # and it improved on where it started


def test_triangulation_inverts_projection():
    rng = np.random.default_rng(9)
    n = 1000
    X = np.stack([rng.uniform(-30, 30, n), rng.uniform(-5, 4, n), rng.uniform(3, 150, n)], 1)
    P, Q = K @ M1, K @ M2
    pl = X @ P[:, :3].T + P[:, 3]; pr = X @ Q[:, :3].T + Q[:, 3]
    got = ops.triangulate_f64(P, Q, dev(pl[:, :2] / pl[:, 2:]), dev(pr[:, :2] / pr[:, 2:])).cpu().numpy()
    assert (np.linalg.norm(got - X, axis=1) / np.linalg.norm(X, axis=1)).max() < 1e-8


@pytest.mark.parametrize("case", ["lds_raised_f64", "beyond_lds_f64", "beyond_lds_f32", "many_pairs_one_wavefront_each"])
def test_refit_kernel_variants_reach_the_same_minimiser(case):
    """The refit has two shapes (four wavefronts per pair for small launches, one for launches that fill the chip) and
    two homes for the supporters (LDS up to a cap that depends on the observation type; beyond it, global memory).  All
    of them must end at the same stationary point: checked against each other through a float64 numpy Gauss-Newton
    polish of the result, which must not move it."""
    rng = np.random.default_rng(77)
    n, stride, n_seg, dtype = {"lds_raised_f64": (3000, 4160, 3, np.float64), "beyond_lds_f64": (4100, 4160, 3, np.float64),
                               "beyond_lds_f32": (4150, 4160, 2, np.float32),
                               "many_pairs_one_wavefront_each": (300, 320, 800, np.float32)}[case]
    T, X, uv_exact = _scene(rng, n)
    uv = (uv_exact + rng.normal(0, 0.5, (n, 2))).astype(dtype)
    start = np.hstack([Rotation.from_rotvec(rng.normal(0, 2e-3, 3)).as_matrix() @ T[:, :3], T[:, 3:] + rng.normal(0, 0.02, (3, 1))])
    Xs = np.zeros((n_seg, stride, 3)); Xs[:, :n] = X
    xy1 = np.zeros((n_seg, stride, 4), dtype); xy1[:, :n, :2] = uv
    sel = np.ones(n, bool); sel[::17] = False
    words = ops.bool_to_mask_words(np.repeat(sel[None], n_seg, 0), stride)
    pose, status = ops.refit_pnp(K, dev(Xs), dev(xy1), dev(np.full(n_seg, n, np.int32)), dev(words), dev(np.repeat(start[None], n_seg, 0)))
    pose, status = pose.cpu().numpy(), status.cpu().numpy()
    assert status.all() and np.abs(pose - pose[0]).max() == 0.0          # every segment: same data, same bits
    got, obs = pose[0], uv.astype(np.float64)[sel]

    def residual(d):
        R = Rotation.from_rotvec(d[:3]).as_matrix() @ got[:, :3]
        P = K @ np.hstack([R, (got[:, 3] + d[3:]).reshape(3, 1)])
        h = X[sel] @ P[:, :3].T + P[:, 3]
        return (h[:, :2] / h[:, 2:] - obs).reshape(-1)

    r0 = residual(np.zeros(6))
    J = np.stack([(residual(np.eye(6)[a] * 1e-6) - residual(-np.eye(6)[a] * 1e-6)) / 2e-6 for a in range(6)], 1)
    step = np.linalg.lstsq(J, -r0, rcond=None)[0]                         # the Gauss-Newton step from the kernel's answer
    assert np.abs(step).max() < 1e-8, step
    assert np.linalg.norm(got - T) / np.linalg.norm(T) < 2e-3

"""Properties of the pieces whose arithmetic lives in cv2 (BFMatcher, solvePnP AP3P / ITERATIVE,
Rodrigues), checked on the ORACLE's restatement against things that do not come from the oracle: scipy's rotation
vectors and matrix exponential, bit-by-bit distances from np.unpackbits, planted poses, and the first-order optimality
condition of the least-squares problem cv2's ITERATIVE solver minimises.  cv2 itself is not available offline, so
which of its tie-breaking and cross-check conventions the reference author's build had stays unpinned (DESIGN.md
section 2); what these tests rule out is the oracle being wrong about the mathematics it restates.  No GPU."""
import numpy as np
import pytest
from scipy.linalg import expm
from scipy.spatial.transform import Rotation

import slam_oracle as orc

K = orc.K


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


@pytest.mark.parametrize("seed", range(4))
def test_rodrigues_is_the_matrix_exponential_and_its_inverse(seed):
    rng = np.random.default_rng(seed)
    for scale in (1e-12, 1e-6, 0.3, 2.0, np.pi - 1e-6):
        r = rng.normal(size=3)
        r *= scale / np.linalg.norm(r)
        R = orc.rodrigues_vec_to_mat(r)
        assert np.allclose(R, expm(_skew(r)), rtol=0, atol=1e-14)
        assert np.allclose(R, Rotation.from_rotvec(r).as_matrix(), rtol=0, atol=1e-14)
        back = orc.rodrigues_mat_to_vec(R).reshape(3)
        if scale < 1e-5:
            assert not back.any()          # cvRodrigues2 returns the zero vector below sin(theta) = 1e-5: an error below 1e-5 rad
        else:
            # compared as rotations: next to pi the vectors r and -(2 pi - |r|) r / |r| are the same rotation, and
            # cvRodrigues2 switches to a diagonal-based formula there (sin(theta) < 1e-5) that is good to ~1e-6
            assert np.allclose(Rotation.from_rotvec(back).as_matrix(), R, rtol=0, atol=1e-5 if scale > 3 else 1e-12)
            if scale <= 3:
                assert np.allclose(back, r, rtol=0, atol=1e-12)
    assert np.array_equal(orc.rodrigues_vec_to_mat(np.zeros(3)), np.eye(3))


def _bit_distance(a, b, cells):
    """Distance between two byte rows from their bits: differing bits (cells = 1) or differing 2-bit cells (cells = 2)."""
    x = np.unpackbits(np.bitwise_xor(a, b))
    return int(x.sum()) if cells == 1 else int(x.reshape(-1, 2).any(axis=1).sum())


@pytest.mark.parametrize("width", [32, 61, 64])
def test_distances_are_the_bit_definitions(width):
    rng = np.random.default_rng(width)
    a = rng.integers(0, 256, (23, width), dtype=np.uint8)
    b = rng.integers(0, 256, (31, width), dtype=np.uint8)
    b[5] = a[7]
    for norm, cells in ((orc.NORM_HAMMING, 1), (orc.NORM_HAMMING2, 2)):
        D = orc.hamming_matrix(a, b, norm)
        want = np.array([[_bit_distance(x, y, cells) for y in b] for x in a])
        assert np.array_equal(D, want) and D[7, 5] == 0


@pytest.mark.parametrize("seed", range(3))
def test_the_two_cross_check_rules_are_what_their_definitions_say(seed):
    """Written from the definitions, not from the oracle's code: (legacy) every train row votes for its first-minimum
    query and a query keeps its best voter, lowest train row on ties; (mutual) a pair survives iff each is the other's
    first-minimum nearest.  Few byte values, so ties abound."""
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 3, (40, 8), dtype=np.uint8)
    t = rng.integers(0, 3, (55, 8), dtype=np.uint8)
    D = np.array([[_bit_distance(x, y, 2) for y in t] for x in q])
    legacy = {}
    for j in range(len(t)):
        i = int(np.argmin(D[:, j]))                       # first minimum
        if i not in legacy or D[i, j] < legacy[i][1]:     # strict: the lower train row stays on equal distance
            legacy[i] = (j, int(D[i, j]))
    gq, gt, gd = orc.bf_match(q, t, orc.NORM_HAMMING2)
    assert list(gq) == sorted(legacy) and [legacy[i] for i in gq] == list(zip(gt.tolist(), gd.tolist()))
    mutual = [(i, int(np.argmin(D[i])), int(D[i].min())) for i in range(len(q)) if int(np.argmin(D[:, int(np.argmin(D[i]))])) == i]
    mq, mt, md = orc.bf_match_mutual(q, t, orc.NORM_HAMMING2)
    assert list(zip(mq.tolist(), mt.tolist(), md.tolist())) == mutual
    assert set(zip(mq.tolist(), mt.tolist())) <= set(zip(gq.tolist(), gt.tolist())) and len(mq) < len(gq)   # a strict subset here


def _scene(rng, n, noise=0.0):
    T = np.hstack([Rotation.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix(), rng.normal(0, 0.5, (3, 1)) + [[0.0], [0.0], [-0.8]]])
    X = np.stack([rng.uniform(-15, 15, n), rng.uniform(-4, 3, n), rng.uniform(5, 60, n)], 1)
    h = X @ (K @ T)[:, :3].T + (K @ T)[:, 3]
    return T, X, h[:, :2] / h[:, 2:] + rng.normal(0, noise, (n, 2))


@pytest.mark.parametrize("seed", range(5))
def test_ap3p_returns_the_planted_pose_on_exact_data(seed):
    """Three exact correspondences admit up to four poses; the fourth point picks one.  With exact data that one is the
    planted pose, whatever the solver's internals."""
    rng = np.random.default_rng(seed)
    T, X, uv = _scene(rng, 400)
    hit = 0
    for k in range(60):
        idx = rng.choice(len(X), 4, replace=False)
        got = orc.pnp(X[idx], uv[idx], K, orc.SOLVEPNP_AP3P)
        if got is None:
            continue
        assert np.linalg.norm(got - T) / np.linalg.norm(T) < 1e-6, (seed, k)
        R = got[:, :3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and abs(np.linalg.det(R) - 1) < 1e-9
        hit += 1
    assert hit >= 55


@pytest.mark.parametrize("seed", range(4))
def test_every_ap3p_candidate_reprojects_its_three_points_and_the_winner_is_best_on_the_fourth(seed):
    rng = np.random.default_rng(100 + seed)
    T, X, uv = _scene(rng, 300, noise=0.4)
    for k in range(25):
        idx = rng.choice(len(X), 4, replace=False)
        with np.errstate(all="ignore"):
            sols = orc.ap3p_solutions(X[idx], uv[idx], K)
        if not sols:
            continue
        errs = []
        for R, t in sols:
            P = K @ np.hstack([R, np.reshape(t, (3, 1))])
            h = X[idx] @ P[:, :3].T + P[:, 3]
            p = h[:, :2] / h[:, 2:]
            assert np.abs(p[:3] - uv[idx[:3]]).max() < 1e-6              # a P3P solution explains its three points
            errs.append(float(np.sum((p[3] - uv[idx[3]]) ** 2)))
        got = orc.pnp(X[idx], uv[idx], K, orc.SOLVEPNP_AP3P)
        assert int(np.argmin(errs)) == 0 and errs == sorted(errs)        # ordered by the fourth point's error; first one returned
        R, t = sols[0]
        U, _, Vt = np.linalg.svd(R)        # solvePnP hands the rotation over as a vector: what comes back is the nearest rotation
        assert np.allclose(got, np.hstack([U @ Vt, np.reshape(t, (3, 1))]), atol=1e-9)


@pytest.mark.parametrize("seed", range(4))
def test_iterative_refit_is_a_stationary_point_of_the_left_reprojection_error(seed):
    """cv2.SOLVEPNP_ITERATIVE minimises the summed squared pixel error; at what it returns the gradient vanishes.  The
    gradient here is a central finite difference over a left-multiplied rotation vector and the translation -- nothing
    of the oracle's Jacobians."""
    rng = np.random.default_rng(200 + seed)
    T, X, uv = _scene(rng, 150, noise=0.5)
    got = orc.pnp(X, uv, K, orc.SOLVEPNP_ITERATIVE)

    def cost(d):
        R = Rotation.from_rotvec(d[:3]).as_matrix() @ got[:, :3]
        P = K @ np.hstack([R, (got[:, 3] + d[3:]).reshape(3, 1)])
        h = X @ P[:, :3].T + P[:, 3]
        return float(np.sum((h[:, :2] / h[:, 2:] - uv) ** 2))

    c0 = cost(np.zeros(6))
    g, curv = np.zeros(6), np.zeros(6)
    for a in range(6):
        e = np.zeros(6)
        e[a] = 1e-5
        g[a] = (cost(e) - cost(-e)) / 2e-5
        curv[a] = (cost(e) - 2 * c0 + cost(-e)) / 1e-10
    # a Newton step along each axis would move the pose by less than the LM stopping step (FLT_EPSILON relative)
    assert (curv > 0).all() and np.abs(g / curv).max() < 1e-6
    assert c0 <= cost(np.r_[np.zeros(3), 1e-4, 0, 0]) and np.linalg.norm(got - T) / np.linalg.norm(T) < 5e-3
    exact = orc.pnp(X, _scene(np.random.default_rng(200 + seed), 150)[2], K, orc.SOLVEPNP_ITERATIVE)
    assert np.linalg.norm(exact - T) / np.linalg.norm(T) < 1e-8          # noise-free: the planted pose

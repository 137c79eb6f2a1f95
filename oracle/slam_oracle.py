"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A float64/numpy restatement of the stereo-trajectory hot path of maoratar1/SLAM
(`utils/utills.py`).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product
(`slam_amd/`) never does and fails loudly when its HIP library is missing.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Two kinds of arithmetic live here:

* PINNED: code that is numpy in the reference (rect test, 4-view join, DLT
  triangulation, projection / consensus test, RANSAC control flow, pose
  composition).  `tests/golden/make_golden.py` imported the reference itself in
  the build container and recorded its outputs; `tests/test_oracle_golden.py`
  holds this module to those vectors bit for bit.
* PARITY UNPINNED: arithmetic that lives inside OpenCV (`cv2.BFMatcher`,
  `cv2.solvePnP` AP3P / ITERATIVE, `cv2.Rodrigues`).  OpenCV is not installed,
  not vendored and not version-pinned by the reference, and the reference has no
  golden vectors for it, so these functions restate the published OpenCV
  algorithms (batch_distance.cpp cross-check, ap3p.cpp Ke & Roumeliotis,
  calibration.cpp DLT + CvLevMarq) from the documented behaviour.  They are the
  definition the HIP path is held to, and are labelled "unpinned" below.
"""
from __future__ import annotations

import cmath
import math

import numpy as np

# cv2 enum values used by the reference (utills.py:24, :503, :563)
NORM_HAMMING = 6
NORM_HAMMING2 = 7
SOLVEPNP_ITERATIVE = 0
SOLVEPNP_AP3P = 5

# utills.py:23,39-41
RATIO = 0.6
PNP_NUM_PTS = 4
SUPP_ERR = 2
REC_ERR = 2

# KITTI seq-00 calibration actually used by the reference (recovered
# from PoseGraphDirectory/pose_graph1.pickle); M2 from the KITTI calib file.
K = np.array([[718.856, 0.0, 607.1928], [0.0, 718.856, 185.2157], [0.0, 0.0, 1.0]])
M1 = np.hstack([np.eye(3), np.zeros((3, 1))])
M2 = np.hstack([np.eye(3), np.array([[-0.5371657188644179], [0.0], [0.0]])])


class DMatch:
    """Duck type of cv2.DMatch: only queryIdx/trainIdx/distance are read by the
    reference (utills.py:69,147,161-162,433,453-454,969)."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, q, t, d, img=0):
        self.queryIdx = int(q)
        self.trainIdx = int(t)
        self.imgIdx = int(img)
        self.distance = float(d)

    def __repr__(self):
        return f"DMatch(q={self.queryIdx}, t={self.trainIdx}, d={self.distance})"


class KeyPoint:
    """Duck type of cv2.KeyPoint: only `.pt` is read (utills.py:163-164)."""
    __slots__ = ("pt",)

    def __init__(self, x, y):
        # cv2 stores float32 coordinates and hands back Python floats
        self.pt = (float(np.float32(x)), float(np.float32(y)))


# --------------------------------------------------------------------------
# Row 1/2: descriptor distances and matching (UNPINNED: cv2.BFMatcher)
# --------------------------------------------------------------------------
_POP1 = np.array([bin(i).count("1") for i in range(256)], dtype=np.int32)
# NORM_HAMMING2: number of non-zero 2-bit cells in a byte (OpenCV popCountTable2)
_POP2 = np.array([sum(1 for c in range(4) if (i >> (2 * c)) & 3) for i in range(256)], dtype=np.int32)


def hamming_matrix(a: np.ndarray, b: np.ndarray, norm: int = NORM_HAMMING2, block: int = 256) -> np.ndarray:
    """All-pairs distance D[i, j] between uint8 rows of a and b (cv2 normHamming
    with cellSize 1 or 2)."""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    lut = _POP2 if norm == NORM_HAMMING2 else _POP1
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.int32)
    for s in range(0, a.shape[0], block):
        x = a[s:s + block, None, :] ^ b[None, :, :]
        out[s:s + block] = lut[x].sum(axis=2)
    return out


def knn2_from_matrix(D: np.ndarray):
    """Two nearest train rows per query; equal distances keep the lower train
    index first (OpenCV batch_distance.cpp insertion: strict `d < dist[K-1]`,
    shift while `dist[k] > d`)."""
    nq, nt = D.shape
    key = D.astype(np.int64) * (nt + 1) + np.arange(nt, dtype=np.int64)[None, :]
    order = np.argsort(key, axis=1, kind="stable")[:, :2]
    dist = np.take_along_axis(D, order, axis=1)
    return order.astype(np.int32), dist.astype(np.int32)


def crosscheck_from_matrix(D: np.ndarray):
    """OpenCV batchDistance(crosscheck=true), K == 1 (utills.py:57-70 ->
    cv2.BFMatcher(crossCheck=True).match).

    For every train row j take its nearest query i*(j) (first minimum).  Scan j
    ascending; query slot i keeps the j with the smallest distance among those
    whose i*(j) == i (strict `<`, so the lowest j wins ties).  Queries that are
    nobody's nearest get no match.  Returns (queryIdx, trainIdx, dist) ascending
    in queryIdx.
    """
    nq, nt = D.shape
    best_d = np.full(nq, np.iinfo(np.int32).max, dtype=np.int64)
    best_j = np.full(nq, -1, dtype=np.int64)
    if nq and nt:
        istar = np.argmin(D, axis=0)            # first minimum per column
        dmin = D[istar, np.arange(nt)]
        for j in range(nt):
            i = istar[j]
            if dmin[j] < best_d[i]:
                best_d[i] = dmin[j]
                best_j[i] = j
    q = np.nonzero(best_j >= 0)[0]
    return q.astype(np.int32), best_j[q].astype(np.int32), best_d[q].astype(np.int32)


def crosscheck_mutual_from_matrix(D: np.ndarray):
    """The OTHER behaviour cv2.BFMatcher(crossCheck=True) has had (newer OpenCV
    batch_distance.cpp computes both directions: tdist/tidx = nearest query of
    every train row, sdist/sidx = nearest train of every query row, and keeps
    train j for query i = tidx[j] only if sidx[i] == j): a pair survives iff each
    row is the other's nearest, first minimum on both sides.  UNPINNED like the
    legacy rule above -- the reference pins no OpenCV version, so both exist
    here and in the HIP matcher (SLAM_MATCH_CROSSCHECK_MUTUAL)."""
    nq, nt = D.shape
    if nq == 0 or nt == 0:
        z = np.zeros(0, np.int32)
        return z, z, z
    sidx = np.argmin(D, axis=1)                 # first minimum per query row
    tidx = np.argmin(D, axis=0)                 # first minimum per train row
    q = np.nonzero(tidx[sidx] == np.arange(nq))[0]
    return q.astype(np.int32), sidx[q].astype(np.int32), D[q, sidx[q]].astype(np.int32)


def bf_match(q_dsc, t_dsc, norm=NORM_HAMMING2):
    """Row 1: `match(..., BRUTE_FORCE)` = cross-check brute force, unsorted
    (utills.py:105-110).  Returns index arrays, not objects."""
    return crosscheck_from_matrix(hamming_matrix(q_dsc, t_dsc, norm))


def bf_match_mutual(q_dsc, t_dsc, norm=NORM_HAMMING2):
    return crosscheck_mutual_from_matrix(hamming_matrix(q_dsc, t_dsc, norm))


def knn_ratio_match(q_dsc, t_dsc, norm=NORM_HAMMING2, ratio=RATIO):
    """Row 2: kNN(k=2) + ratio test `first.distance < ratio*second.distance`
    evaluated in float64 (utills.py:92-102, :131-152)."""
    idx, dist = knn2_from_matrix(hamming_matrix(q_dsc, t_dsc, norm))
    d1 = dist[:, 0].astype(np.float64)
    d2 = dist[:, 1].astype(np.float64)
    keep = d1 < ratio * d2
    q = np.nonzero(keep)[0].astype(np.int32)
    return q, idx[q, 0], dist[q, 0]


def to_dmatches(q, t, d):
    out = np.empty(len(q), dtype=object)
    for k in range(len(q)):
        out[k] = DMatch(q[k], t[k], d[k])
    return out


# --------------------------------------------------------------------------
# Row 3/4: rectified-stereo rejection and the 4-view join (PINNED)
# --------------------------------------------------------------------------
def rect_filter(y_left, y_right, q, t, rec_err=REC_ERR):
    """utills.py:168-183: keep match k iff |y_right - y_left| <= REC_ERR
    (inclusive) in float64.  Returns indices into the match list."""
    yl = np.asarray(y_left, dtype=np.float64)[q]
    yr = np.asarray(y_right, dtype=np.float64)[t]
    ok = np.abs(yr - yl) <= rec_err
    return np.nonzero(ok)[0], np.nonzero(~ok)[0]


def join_4view(q0, rec0_idx, q1, rec1_idx, tq, tt):
    """utills.py:423-471: rec{0,1} map left key-point -> index in the pair's
    match list for rect-passing matches; walk the left0->left1 matches in order
    and keep those whose ends are both in the maps.  Returns (idx into pair0
    matches, idx into pair1 matches, idx into temporal matches)."""
    rec0 = {int(q0[k]): int(k) for k in rec0_idx}
    rec1 = {int(q1[k]): int(k) for k in rec1_idx}
    a, b, c = [], [], []
    for k in range(len(tq)):
        l0, l1 = int(tq[k]), int(tt[k])
        if l0 in rec0 and l1 in rec1:
            a.append(rec0[l0])
            b.append(rec1[l1])
            c.append(k)
    return np.array(a, dtype=np.int64), np.array(b, dtype=np.int64), np.array(c, dtype=np.int64)


# --------------------------------------------------------------------------
# Row 6: DLT triangulation (PINNED: numpy/LAPACK in the reference)
# --------------------------------------------------------------------------
def triangulate(P, Q, xy_l, xy_r):
    """utills.py:220-255: per correspondence the 4x4 DLT matrix, np.linalg.svd,
    last row of Vt, dehomogenise."""
    P = np.asarray(P, dtype=np.float64)
    Q = np.asarray(Q, dtype=np.float64)
    xy_l = np.asarray(xy_l, dtype=np.float64).reshape(-1, 2)
    xy_r = np.asarray(xy_r, dtype=np.float64).reshape(-1, 2)
    out = np.empty((len(xy_l), 3))
    for i in range(len(xy_l)):
        A = np.array([xy_l[i, 0] * P[2] - P[0],
                      xy_l[i, 1] * P[2] - P[1],
                      xy_r[i, 0] * Q[2] - Q[0],
                      xy_r[i, 1] * Q[2] - Q[1]])
        _, _, vh = np.linalg.svd(A, compute_uv=True)
        v = vh[-1]
        out[i] = v[:3] / v[3]
    return out


# --------------------------------------------------------------------------
# Row 8: composition, projection, consensus test (PINNED)
# --------------------------------------------------------------------------
def compose(first, second):
    """utills.py:659-669: second @ [first; 0 0 0 1]."""
    hom = np.append(first, [np.array([0, 0, 0, 1])], axis=0)
    return second @ hom


def project(X, P):
    """utills.py:698-705."""
    hom = X @ P[:, :3].T + P[:, 3].T
    return hom[:, :2] / hom[:, [-1]]


def check_projection(proj, obs, acc=SUPP_ERR):
    """utills.py:708-736 with the default dim="3d": the UN-squared Euclidean norm
    is compared with acc**2, i.e. an inclusive 4.0 px radius."""
    return np.linalg.norm(proj - obs, axis=1) <= acc ** 2


def consensus_match(calib, T, M2_, X, left1, right1, acc=SUPP_ERR):
    """utills.py:533-545, :672-695: only left1 and right1 are tested."""
    Tr = compose(T, M2_)
    Pl = calib @ T
    Pr = calib @ Tr
    return check_projection(project(X, Pl), left1, acc) & check_projection(project(X, Pr), right1, acc)


# --------------------------------------------------------------------------
# cv2.Rodrigues (UNPINNED: OpenCV calibration.cpp cvRodrigues2)
# --------------------------------------------------------------------------
def rodrigues_vec_to_mat(r):
    r = np.asarray(r, dtype=np.float64).reshape(3)
    theta = math.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = math.cos(theta), math.sin(theta)
    c1 = 1.0 - c
    k = r / theta
    rrt = np.outer(k, k)
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + c1 * rrt + s * kx


def rodrigues_mat_to_vec(R):
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    # checkRange(R, true, NULL, -100, 100): NaN / out-of-range -> zero vector
    if not np.all(np.isfinite(R)) or np.any(R < -100) or np.any(R >= 100):
        return np.zeros(3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    rx, ry, rz = R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]
    s = math.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5
    c = 1.0 if c > 1.0 else (-1.0 if c < -1.0 else c)
    theta = math.acos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        rx = math.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        ry = math.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        rz = math.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        theta /= math.sqrt(rx * rx + ry * ry + rz * rz)
        return np.array([rx, ry, rz]) * theta
    vth = 1.0 / (2.0 * s)
    vth *= theta
    return np.array([rx, ry, rz]) * vth


# --------------------------------------------------------------------------
# Row 7: AP3P minimal solver (UNPINNED: OpenCV ap3p.cpp, Ke & Roumeliotis 2017)
# --------------------------------------------------------------------------
def _solve_quartic(f):
    """ap3p.cpp solveQuartic (Ferrari), IEEE semantics via numpy scalars."""
    a4, a3, a2, a1, a0 = (np.float64(v) for v in f)
    a4_2 = a4 * a4
    a3_2 = a3 * a3
    a4_3 = a4_2 * a4
    a2a4 = a2 * a4
    p4 = (8 * a2a4 - 3 * a3_2) / (8 * a4_2)
    q4 = (a3_2 * a3 - 4 * a2a4 * a3 + 8 * a1 * a4_2) / (8 * a4_3)
    r4 = (256 * a0 * a4_3 - 3 * (a3_2 * a3_2) - 64 * a1 * a3 * a4_2 + 16 * a2a4 * a3_2) / (256 * (a4_3 * a4))
    p3 = ((p4 * p4) / 12 + r4) / 3
    q3 = (72 * r4 * p4 - 2 * p4 * p4 * p4 - 27 * q4 * q4) / 432
    disc = np.complex128(q3 * q3 - p3 * p3 * p3)
    if q3 >= 0:
        w = -np.sqrt(disc) - q3
    else:
        w = np.sqrt(disc) - q3
    if w.imag == 0.0:
        wr = np.cbrt(w.real)
        t = 2.0 * (wr + p3 / wr)
    else:
        w = np.power(w, 1.0 / 3)
        t = 4.0 * w.real
    sqrt_2m = np.sqrt(np.complex128(-2 * p4 / 3 + t))
    B_4A = -a3 / (4 * a4)
    complex1 = 4 * p4 / 3 + t
    complex2 = 2 * q4 / sqrt_2m
    sqrt_2m_rh = sqrt_2m.real / 2
    sqrt1 = np.sqrt(-(complex1 + complex2)).real / 2
    sqrt2 = np.sqrt(-(complex1 - complex2)).real / 2
    return [B_4A + sqrt_2m_rh + sqrt1, B_4A + sqrt_2m_rh - sqrt1,
            B_4A - sqrt_2m_rh + sqrt2, B_4A - sqrt_2m_rh - sqrt2]


def _polish_quartic(c, roots):
    """ap3p.cpp polishQuarticRoots: two Newton sweeps."""
    for _ in range(2):
        for j in range(4):
            r = roots[j]
            err = (((c[0] * r + c[1]) * r + c[2]) * r + c[3]) * r + c[4]
            der = ((4 * c[0] * r + 3 * c[1]) * r + 2 * c[2]) * r + c[3]
            roots[j] = r - err / der
    return roots


def _vnorm(v):
    return np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])


def ap3p_solutions(obj, img, calib):
    """All AP3P poses from the first three correspondences, ordered by the 4th
    point's reprojection error (stable insertion sort on strict `>`, the p4p
    path of ap3p.cpp computePoses).  Returns a list of (R, t).  Degenerate input
    (repeated points) propagates NaN exactly as the C++ does: the roots are NaN,
    `abs(NaN) > 1` is false, so four NaN poses come back."""
    obj = np.asarray(obj, dtype=np.float64)
    img = np.asarray(img, dtype=np.float64)
    fx, fy, cx, cy = (np.float64(v) for v in (calib[0, 0], calib[1, 1], calib[0, 2], calib[1, 2]))
    with np.errstate(all="ignore"):
        inv_fx, inv_fy, cx_fx, cy_fy = 1.0 / fx, 1.0 / fy, cx / fx, cy / fy
        b = []
        for i in range(3):
            mu = inv_fx * img[i, 0] - cx_fx
            mv = inv_fy * img[i, 1] - cy_fy
            mk = 1.0 / np.sqrt(mu * mu + mv * mv + 1)
            b.append(np.array([mu * mk, mv * mk, mk]))
        mu3 = inv_fx * img[3, 0] - cx_fx
        mv3 = inv_fy * img[3, 1] - cy_fy
        w1, w2, w3, w4 = obj[0], obj[1], obj[2], obj[3]
        b1, b2, b3 = b
        u0 = w1 - w2
        nu0 = _vnorm(u0)
        k1 = u0 / nu0
        k3 = np.cross(b1, b2)
        nk3 = _vnorm(k3)
        k3 = k3 / nk3
        tz = np.cross(b1, k3)
        v1 = np.cross(b1, b3)
        v2 = np.cross(b2, b3)
        u1 = w1 - w3
        u1k1 = u1 @ k1
        k3b3 = k3 @ b3
        f11 = k3b3
        f13 = k3 @ v1
        f15 = -u1k1 * f11
        nl = np.cross(u1, k1)
        delta = _vnorm(nl)
        nl = nl / delta
        f11 = f11 * delta
        f13 = f13 * delta
        u2k1 = u1k1 - nu0
        f21 = tz @ v2
        f22 = nk3 * k3b3
        f23 = k3 @ v2
        f24 = u2k1 * f22
        f25 = -u2k1 * f21
        f21 = f21 * delta
        f22 = f22 * delta
        f23 = f23 * delta
        g1 = f13 * f22
        g2 = f13 * f25 - f15 * f23
        g3 = f11 * f23 - f13 * f21
        g4 = -f13 * f24
        g5 = f11 * f22
        g6 = f11 * f25 - f15 * f21
        g7 = -f15 * f24
        coeffs = [g5 * g5 + g1 * g1 + g3 * g3,
                  2 * (g5 * g6 + g1 * g2 + g3 * g4),
                  g6 * g6 + 2 * g5 * g7 + g2 * g2 + g4 * g4 - g1 * g1 - g3 * g3,
                  2 * (g6 * g7 - g1 * g2 - g3 * g4),
                  g7 * g7 - g2 * g2 - g4 * g4]
        s = _polish_quartic(coeffs, _solve_quartic(coeffs))
        temp = np.cross(k1, nl)
        Ck1nl = np.array([[k1[0], nl[0], temp[0]], [k1[1], nl[1], temp[1]], [k1[2], nl[2], temp[2]]])
        Cb1k3tzT = np.array([b1, k3, tz])
        b3p = (delta / k3b3) * b3
        sols, errs = [], []
        for i in range(4):
            ct1 = s[i]
            if abs(ct1) > 1:          # NaN falls through, as in C++
                continue
            st1 = np.sqrt(1 - ct1 * ct1)
            st1 = st1 if k3b3 > 0 else -st1
            ct3 = g1 * ct1 + g2
            st3 = g3 * ct1 + g4
            nt3 = st1 / ((g5 * ct1 + g6) * ct1 + g7)
            ct3 = ct3 * nt3
            st3 = st3 * nt3
            C13 = np.array([[ct3, 0, -st3],
                            [st1 * st3, ct1, st1 * ct3],
                            [ct1 * st3, -st1, ct1 * ct3]])
            R = (Ck1nl @ C13) @ Cb1k3tzT
            rp3 = R.T @ w3
            t = st1 * b3p - rp3
            Rsol = R.T.copy()
            Xp = Rsol[0] @ w4 + t[0]
            Yp = Rsol[1] @ w4 + t[1]
            Zp = Rsol[2] @ w4 + t[2]
            mu3p = Xp / Zp
            mv3p = Yp / Zp
            e = (mu3p - mu3) * (mu3p - mu3) + (mv3p - mv3) * (mv3p - mv3)
            sols.append((Rsol, np.array(t)))
            errs.append(e)
        for i in range(1, len(sols)):
            j = i
            while j > 0 and errs[j - 1] > errs[j]:
                errs[j], errs[j - 1] = errs[j - 1], errs[j]
                sols[j], sols[j - 1] = sols[j - 1], sols[j]
                j -= 1
    return sols


def solve_pnp_ap3p(obj, img, calib):
    """cv2.solvePnP(obj[4,3], img[4,2], K, None, flags=AP3P) -> (ok, rvec, tvec).
    ok is False only when AP3P yields no root in [-1, 1]."""
    sols = ap3p_solutions(np.asarray(obj, dtype=np.float64), np.asarray(img, dtype=np.float64), calib)
    if not sols:
        return False, None, None
    R, t = sols[0]
    return True, rodrigues_mat_to_vec(R).reshape(3, 1), t.reshape(3, 1)


# --------------------------------------------------------------------------
# Row 10: iterative PnP (UNPINNED: cvFindExtrinsicCameraParams2 + CvLevMarq)
# --------------------------------------------------------------------------
def _project_with_jac(obj, r, t, calib, want_jac):
    fx, fy, cx, cy = calib[0, 0], calib[1, 1], calib[0, 2], calib[1, 2]
    R = rodrigues_vec_to_mat(r)
    Xc = obj @ R.T + t
    iz = 1.0 / Xc[:, 2]
    x = Xc[:, 0] * iz
    y = Xc[:, 1] * iz
    uv = np.stack([fx * x + cx, fy * y + cy], axis=1)
    if not want_jac:
        return uv, None
    n = len(obj)
    J = np.zeros((2 * n, 6))
    # d(uv)/d(Xc)
    du = np.stack([fx * iz, np.zeros(n), -fx * x * iz], axis=1)
    dv = np.stack([np.zeros(n), fy * iz, -fy * y * iz], axis=1)
    # d(R X)/d r = -[R X]x J_l(r)   (exact derivative of the Rodrigues map)
    th = math.sqrt(float(r @ r))
    rx = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    if th < 1e-8:
        Jl = np.eye(3) + 0.5 * rx
    else:
        Jl = np.eye(3) + (1 - math.cos(th)) / (th * th) * rx + (th - math.sin(th)) / (th ** 3) * (rx @ rx)
    RX = Xc - t
    for i in range(n):
        a = RX[i]
        ax = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        dXdr = -ax @ Jl
        J[2 * i, :3] = du[i] @ dXdr
        J[2 * i + 1, :3] = dv[i] @ dXdr
        J[2 * i, 3:] = du[i]
        J[2 * i + 1, 3:] = dv[i]
    return uv, J


def solve_pnp_iterative(obj, img, calib, max_iter=20, eps=float(np.finfo(np.float32).eps)):
    """cv2.solvePnP(..., flags=SOLVEPNP_ITERATIVE) for a non-planar point set:
    DLT initialisation on K-normalised points, then Levenberg-Marquardt on the
    pixel reprojection error in (rvec, tvec) with OpenCV's CvLevMarq schedule
    (lambda = 10**-3, x10 on a worse step, /10 on a better one; stop after 20
    iterations or a relative parameter change below FLT_EPSILON)."""
    obj = np.asarray(obj, dtype=np.float64).reshape(-1, 3)
    img = np.asarray(img, dtype=np.float64).reshape(-1, 2)
    n = len(obj)
    if n < 6:
        return False, None, None
    fx, fy, cx, cy = calib[0, 0], calib[1, 1], calib[0, 2], calib[1, 2]
    mn = np.stack([(img[:, 0] - cx) / fx, (img[:, 1] - cy) / fy], axis=1)
    L = np.zeros((2 * n, 12))
    L[0::2, 0:3] = obj
    L[0::2, 3] = 1
    L[0::2, 8:11] = -mn[:, [0]] * obj
    L[0::2, 11] = -mn[:, 0]
    L[1::2, 4:7] = obj
    L[1::2, 7] = 1
    L[1::2, 8:11] = -mn[:, [1]] * obj
    L[1::2, 11] = -mn[:, 1]
    _, _, Vt = np.linalg.svd(L.T @ L)
    RRt = Vt[11].reshape(3, 4).copy()
    if np.linalg.det(RRt[:, :3]) < 0:
        RRt = -RRt
    sc = np.linalg.norm(RRt[:, :3])
    U, _, Vt2 = np.linalg.svd(RRt[:, :3])
    R = U @ Vt2
    t = RRt[:, 3] * (np.linalg.norm(R) / sc)
    r = rodrigues_mat_to_vec(R)

    param = np.concatenate([r, t])
    lam_lg10 = -3
    iters = 0
    uv, J = _project_with_jac(obj, param[:3], param[3:], calib, True)
    err = (uv - img).reshape(-1)
    prev_err_norm = None
    while True:
        JtJ = J.T @ J
        JtErr = J.T @ err
        prev_param = param.copy()
        if prev_err_norm is None:
            prev_err_norm = np.linalg.norm(err)
        while True:
            lam = math.exp(lam_lg10 * math.log(10.0))
            A = JtJ.copy()
            A[np.diag_indices(6)] *= 1.0 + lam
            step = np.linalg.lstsq(A, JtErr, rcond=None)[0]
            param = prev_param - step
            uv, _ = _project_with_jac(obj, param[:3], param[3:], calib, False)
            err = (uv - img).reshape(-1)
            err_norm = np.linalg.norm(err)
            if err_norm > prev_err_norm:
                lam_lg10 += 1
                if lam_lg10 <= 16:
                    continue
            break
        lam_lg10 = max(lam_lg10 - 1, -16)
        iters += 1
        rel = np.linalg.norm(param - prev_param) / max(np.linalg.norm(prev_param), 1e-300)
        if iters >= max_iter or rel < eps:
            break
        prev_err_norm = err_norm
        uv, J = _project_with_jac(obj, param[:3], param[3:], calib, True)
        err = (uv - img).reshape(-1)
    return True, param[:3].reshape(3, 1), param[3:].reshape(3, 1)


def pnp(obj, img, calib, flag):
    """utills.py:585-596 + :574-582."""
    if flag == SOLVEPNP_AP3P:
        ok, rvec, tvec = solve_pnp_ap3p(obj, img, calib)
    else:
        ok, rvec, tvec = solve_pnp_iterative(obj, img, calib)
    if not ok:
        return None
    return np.hstack((rodrigues_vec_to_mat(rvec), tvec.reshape(3, 1)))


# --------------------------------------------------------------------------
# Row 9/10: adaptive RANSAC (PINNED control flow; draws from np.random)
# --------------------------------------------------------------------------
def ransac_iter_estimate(prob, outliers_perc, model_parm_num=PNP_NUM_PTS):
    """utills.py:488-489."""
    with np.errstate(divide="ignore"):
        return np.log(1 - prob) / np.log(1 - np.power(1 - outliers_perc, model_parm_num))


def extra_refine(mask, X, left1, M2_, right1, calib=K, acc=SUPP_ERR):
    """utills.py:548-571 with it_num=1."""
    if mask is None or int(np.sum(mask)) < 6:
        return None, [False] * (0 if mask is None else len(mask))
    T = pnp(X[mask], left1[mask], calib, SOLVEPNP_ITERATIVE)
    new_mask = [False] * len(mask)
    if T is not None:
        new_mask = consensus_match(calib, T, M2_, X, left1, right1, acc)
    return T, new_mask


def online_est_pnp_ransac(X, left1, M2_, right1, calib=K, acc=SUPP_ERR, model_parm_num=PNP_NUM_PTS,
                          prob=0.9999, max_iterations=1e3, max_draws=None, fixed_iterations=None, stats=None):
    """utills.py:474-530.  Draws come from the global np.random stream exactly as
    in the reference.  `max_draws` bounds the reference's unbounded retry loop
    (utills.py:506-507); `fixed_iterations` disables the adaptive stop (stress
    config).  Returns (refit pose | None, PRE-refit best mask)."""
    n = len(left1)
    if n == 0:
        raise ValueError("low >= high")          # np.random.randint(0, 0, 4)
    max_supp, best_mask, best_draw = -1, None, -1
    inl, outl = 0, 0
    it, draws = 0, 0
    outliers_perc = 0.99
    while True:
        if fixed_iterations is None:
            if not (outliers_perc != 0 and it < ransac_iter_estimate(prob, outliers_perc, model_parm_num)
                    and it < max_iterations):
                break
        elif it >= fixed_iterations:
            break
        if max_draws is not None and draws >= max_draws:
            break
        idx = np.random.randint(0, n, model_parm_num)
        draws += 1
        T = pnp(X[idx], left1[idx], calib, SOLVEPNP_AP3P)
        if T is None:
            continue
        mask = consensus_match(calib, T, M2_, X, left1, right1, acc)
        num = int(np.sum(mask))
        if num > max_supp:
            max_supp, best_mask, best_draw = num, mask, draws - 1
        it += 1
        outl += n - num
        inl += num
        outliers_perc = min(outl / (inl + outl), 0.99)
    if stats is not None:
        stats.update(iterations=it, draws=draws, best_draw=best_draw, best_count=max_supp)
    if best_mask is None:
        # reference: sum(None) raises TypeError in extra_refine
        raise TypeError("'NoneType' object is not iterable")
    T, _ = extra_refine(best_mask, X, left1, M2_, right1, calib, acc)
    return T, best_mask


# --------------------------------------------------------------------------
# Row 11: per-pair driver and sequence composition
# --------------------------------------------------------------------------
def stereo_match(dsc_l, dsc_r, xy_l, xy_r, norm=NORM_HAMMING2, matcher=None):
    """read_and_rec_match minus detection (utills.py:386-396)."""
    q, t, d = (matcher or bf_match)(dsc_l, dsc_r, norm)
    rec_idx, _ = rect_filter(xy_l[:, 1], xy_r[:, 1], q, t)
    return q, t, d, rec_idx


def track_pair(frame0, frame1, calib=K, M1_=M1, M2_=M2, norm=NORM_HAMMING2, stats=None, matcher=None, **ransac_kw):
    """find_features_in_consecutive_frames (utills.py:876-925) on index arrays.
    frame = dict(dsc_l, dsc_r, xy_l, xy_r[, stereo]).  Returns dict with the
    tracked key-point indices (L0, R0, L1, R1), points, pose, mask, inlier %.
    matcher: what match() does (utills.py:105-119); default = bf_match (BRUTE_FORCE),
    or bf_match_mutual / knn_ratio_match."""
    matcher = matcher or bf_match
    s0 = frame0.get("stereo") or stereo_match(frame0["dsc_l"], frame0["dsc_r"], frame0["xy_l"], frame0["xy_r"], norm, matcher)
    s1 = frame1.get("stereo") or stereo_match(frame1["dsc_l"], frame1["dsc_r"], frame1["xy_l"], frame1["xy_r"], norm, matcher)
    q0, t0, _, rec0 = s0
    q1, t1, _, rec1 = s1
    tq, tt, _ = matcher(frame0["dsc_l"], frame1["dsc_l"], norm)
    a, b, _ = join_4view(q0, rec0, q1, rec1, tq, tt)
    kp = np.stack([q0[a], t0[a], q1[b], t1[b]], axis=1).astype(np.int32) if len(a) else np.zeros((0, 4), np.int32)
    l0 = frame0["xy_l"][kp[:, 0]].astype(np.float64)
    r0 = frame0["xy_r"][kp[:, 1]].astype(np.float64)
    l1 = frame1["xy_l"][kp[:, 2]].astype(np.float64)
    r1 = frame1["xy_r"][kp[:, 3]].astype(np.float64)
    X = triangulate(calib @ M1_, calib @ M2_, l0, r0)
    T, mask = online_est_pnp_ransac(X, l1, M2_, r1, calib, stats=stats, **ransac_kw)
    mask = np.asarray(mask, dtype=bool)
    return dict(kp=kp, X=X, l0=l0, r0=r0, l1=l1, r1=r1, T=T, mask=mask,
                inliers_perc=100 * int(mask.sum()) / len(mask))


def rel_to_global(T_arr):
    """utills.py:749-757."""
    out, last = [], T_arr[0]
    for t in T_arr:
        last = compose(last, t)
        out.append(last)
    return out


def camera_centres(T_arr):
    """utills.py:599-606, :785-794: -R^T t per pose."""
    return np.array([-1 * t[:, :3].T @ t[:, 3] for t in T_arr])

"""Pinning harness for the arithmetic the reference delegates to OpenCV (cv2.BFMatcher cross-check and
kNN, cv2.solvePnP AP3P / ITERATIVE, cv2.Rodrigues -- utils/utills.py:65-66, :100-101, :592, :581).

cv2 is not installed in the build image or on the GPU box, so these conventions are "parity unpinned" there.  This
module makes the comparison a one-liner wherever cv2 IS installed: the same seeded inputs go through `cv2` exactly as
the reference calls it and through a SUBJECT (the CPU oracle, or the HIP path behind slam_amd.utills), and the report
says which cross-check rule the installed OpenCV follows and how far the poses are apart.  tests/test_cv2_pin.py runs it
(a) always, with an oracle-backed stand-in for cv2, so the harness itself is exercised, and (b) against the real cv2
whenever `import cv2` succeeds."""
import numpy as np

K = np.array([[718.856, 0.0, 607.1928], [0.0, 718.856, 185.2157], [0.0, 0.0, 1.0]])
RULES = ("ocv_legacy", "mutual")


# ---- subjects -------------------------------------------------------------------------------------------------------
class OracleSubject:
    """The CPU restatement (oracle/slam_oracle.py)."""
    name = "oracle"

    def __init__(self):
        import slam_oracle as orc
        self.orc = orc

    def cross_check(self, norm, q, t, rule):
        f = self.orc.bf_match_mutual if rule == "mutual" else self.orc.bf_match
        qi, ti, d = f(q, t, norm)
        return [(int(a), int(b), int(c)) for a, b, c in zip(qi, ti, d)]

    def knn2(self, norm, q, t):
        idx, dist = self.orc.knn2_from_matrix(self.orc.hamming_matrix(q, t, norm))
        return [[(int(idx[i, k]), int(dist[i, k])) for k in range(idx.shape[1])] for i in range(len(q))]

    def pnp(self, X, uv, flag):
        with np.errstate(all="ignore"):
            return self.orc.pnp(X, uv, K, flag)

    def rodrigues(self, r):
        return self.orc.rodrigues_vec_to_mat(r)


class HipSubject:
    """The reference-shaped entry points of the HIP path (slam_amd/utills.py)."""
    name = "hip"

    def __init__(self):
        from slam_amd import utills
        self.u = utills

    def cross_check(self, norm, q, t, rule):
        saved = self.u.CROSSCHECK_RULE
        self.u.CROSSCHECK_RULE = rule
        try:
            return [(m.queryIdx, m.trainIdx, int(m.distance)) for m in self.u.bf_matching(norm, q, t, sort=False)]
        finally:
            self.u.CROSSCHECK_RULE = saved

    def knn2(self, norm, q, t):
        return [[(m.trainIdx, int(m.distance)) for m in row] for row in self.u.knn_matching(norm, q, t)]

    def pnp(self, X, uv, flag):
        return self.u.pnp(X, uv, K, flag)

    def rodrigues(self, r):
        return self.u.rodriguez_to_mat(r, np.zeros(3))[:, :3]


# ---- inputs ---------------------------------------------------------------------------------------------------------
def _descriptors(rng, n, width, values):
    d = rng.integers(0, values, (n, width), dtype=np.uint8)
    if n > 6:
        d[rng.integers(0, n, n // 4)] = d[rng.integers(0, n, n // 4)]        # exact duplicates
    return d


def _matcher_cases(rng):
    out = []
    for nq, nt, width, values in ((1, 1, 61, 256), (37, 5, 61, 256), (5, 37, 32, 256), (300, 420, 61, 256), (420, 300, 61, 4),
                                  (257, 255, 32, 2), (640, 640, 61, 256), (130, 700, 64, 3)):
        q, t = _descriptors(rng, nq, width, values), _descriptors(rng, nt, width, values)
        m = min(nq, nt) // 2
        if m > 10:
            t[:m] = q[:m]                                                      # true matches ...
            if values == 256:                                                  # ... with ~6 % of their bits flipped
                flips = rng.integers(0, 256, (4,) + t[:m].shape, dtype=np.uint8)
                t[:m] ^= flips[0] & flips[1] & flips[2] & flips[3]
        out.append((q, t, "ties" if values < 256 else "random"))
    return out


def _scene(rng, n, noise):
    th = rng.normal(0, 0.03, 3)
    a = np.linalg.norm(th)
    k = th / a
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0.0]])
    R = np.cos(a) * np.eye(3) + (1 - np.cos(a)) * np.outer(k, k) + np.sin(a) * kx
    T = np.hstack([R, rng.normal(0, 0.3, (3, 1)) + [[0.0], [0.0], [-0.8]]])
    X = np.stack([rng.uniform(-15, 15, n), rng.uniform(-4, 3, n), rng.uniform(5, 60, n)], 1)
    h = X @ (K @ T)[:, :3].T + (K @ T)[:, 3]
    return T, X, h[:, :2] / h[:, 2:] + rng.normal(0, noise, (n, 2))


def _cv2_pose(cv2, X, uv, flag):
    """As utils/utills.py:585-596 does it: solvePnP, then Rodrigues on the rotation vector."""
    ok, rvec, tvec = cv2.solvePnP(np.ascontiguousarray(X, dtype=np.float64), np.ascontiguousarray(uv, dtype=np.float64), K, None, flags=flag)
    if not ok:
        return None
    R = cv2.Rodrigues(rvec)[0]
    return np.hstack([np.asarray(R, dtype=np.float64), np.asarray(tvec, dtype=np.float64).reshape(3, 1)])


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


# ---- the comparison ---------------------------------------------------------------------------------------------------
def report(cv2, subject, seed=11, n_ap3p=200, n_refit=12):
    rng = np.random.default_rng(seed)
    rep = {"subject": subject.name, "cv2": getattr(cv2, "__version__", "stand-in")}

    # cross-check matcher: which rule does this cv2 follow, ties included
    cases = _matcher_cases(rng)
    for norm_name in ("NORM_HAMMING", "NORM_HAMMING2"):
        norm = getattr(cv2, norm_name)
        agree = {r: 0 for r in RULES}
        order_ok = 0
        knn_agree = {"random": 0, "ties": 0}
        knn_cases = {"random": 0, "ties": 0}
        for q, t, kind in cases:
            want = [(m.queryIdx, m.trainIdx, int(round(m.distance))) for m in cv2.BFMatcher(norm, crossCheck=True).match(q, t)]
            order_ok += [w[0] for w in want] == sorted(w[0] for w in want)
            for rule in RULES:
                agree[rule] += sorted(subject.cross_check(norm, q, t, rule)) == sorted(want)
            rows = cv2.BFMatcher(norm).knnMatch(q, t, 2)
            want_knn = [[(m.trainIdx, int(round(m.distance))) for m in row] for row in rows]
            knn_cases[kind] += 1
            knn_agree[kind] += subject.knn2(norm, q, t) == want_knn
        rep[norm_name] = {"cases": len(cases), "cross_check_equal": agree, "cv2_returns_query_order": order_ok == len(cases),
                          "knn2_equal": knn_agree, "knn2_cases": knn_cases}

    # Rodrigues
    worst = 0.0
    for scale in (1e-9, 1e-4, 0.02, 0.7, 2.5):
        r = rng.normal(size=3)
        r *= scale / np.linalg.norm(r)
        worst = max(worst, float(np.abs(np.asarray(cv2.Rodrigues(r.reshape(3, 1))[0]) - subject.rodrigues(r)).max()))
    rep["rodrigues_max_abs_diff"] = worst

    # AP3P on minimal samples (noisy data: up to four roots, the fourth point picks)
    T, X, uv = _scene(rng, 400, 0.4)
    both = same = one_sided = 0
    worst_ap3p = []
    for _ in range(n_ap3p):
        idx = rng.choice(len(X), 4, replace=False)
        want, got = _cv2_pose(cv2, X[idx], uv[idx], cv2.SOLVEPNP_AP3P), subject.pnp(X[idx], uv[idx], cv2.SOLVEPNP_AP3P)
        if (want is None) != (got is None):
            one_sided += 1
        elif want is not None:
            both += 1
            e = _rel(got, want)
            same += e <= 1e-6
            worst_ap3p.append(e)
    rep["ap3p"] = {"samples": n_ap3p, "both_solved": both, "within_1e-6": int(same), "solved_by_one_only": one_sided,
                   "median_rel": float(np.median(worst_ap3p)) if worst_ap3p else None}

    # ITERATIVE refit on supporter-sized sets
    errs = []
    for k in range(n_refit):
        T, X, uv = _scene(rng, 60 + 40 * k, 0.5)
        want, got = _cv2_pose(cv2, X, uv, cv2.SOLVEPNP_ITERATIVE), subject.pnp(X, uv, cv2.SOLVEPNP_ITERATIVE)
        errs.append(np.inf if want is None or got is None else _rel(got, want))
    rep["iterative"] = {"sets": n_refit, "within_1e-5": int(np.sum(np.array(errs) <= 1e-5)), "worst_rel": float(np.max(errs))}
    return rep


def verdict(rep):
    """What a report pins: (rule the installed cv2 follows per norm | None, list of findings)."""
    findings = []
    rules = {}
    for norm_name in ("NORM_HAMMING", "NORM_HAMMING2"):
        r = rep[norm_name]
        full = [rule for rule in RULES if r["cross_check_equal"][rule] == r["cases"]]
        rules[norm_name] = full[0] if full else None
        if not full:
            findings.append(f"{norm_name}: neither cross-check rule reproduces cv2 on all cases {r['cross_check_equal']}")
        for kind in ("random", "ties"):
            if r["knn2_equal"][kind] != r["knn2_cases"][kind]:
                findings.append(f"{norm_name}: kNN lists differ on {kind} data ({r['knn2_equal'][kind]}/{r['knn2_cases'][kind]})")
    if rep["rodrigues_max_abs_diff"] > 1e-12:
        findings.append(f"Rodrigues differs by {rep['rodrigues_max_abs_diff']:.2e}")
    a = rep["ap3p"]
    if a["solved_by_one_only"] > 0.02 * a["samples"] or a["within_1e-6"] < 0.95 * a["both_solved"]:
        findings.append(f"AP3P: {a}")
    if rep["iterative"]["within_1e-5"] != rep["iterative"]["sets"]:
        findings.append(f"ITERATIVE refit: {rep['iterative']}")
    return rules, findings

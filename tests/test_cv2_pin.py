"""The OpenCV conventions the reference inherits (cross-check rule, kNN tie order, AP3P root choice, ITERATIVE refit,
Rodrigues) against an installed cv2 -- skipped where `import cv2` fails, as in the build image and on the GPU box, which
is why DESIGN.md section 2 calls them unpinned -- and, always, the same harness (tests/cv2_pin.py) against a stand-in
for cv2 made of the oracle, so that the comparison code itself is known to run: with cv2 present these tests turn
"unpinned" into a report without further work."""
import json
import types

import numpy as np
import pytest

import cv2_pin
import slam_oracle as orc


def _standin(mutual=False):
    cv2 = types.SimpleNamespace(NORM_HAMMING=orc.NORM_HAMMING, NORM_HAMMING2=orc.NORM_HAMMING2,
                                SOLVEPNP_ITERATIVE=orc.SOLVEPNP_ITERATIVE, SOLVEPNP_AP3P=orc.SOLVEPNP_AP3P)

    class BFMatcher:
        def __init__(self, norm, crossCheck=False):
            self.norm, self.cross = norm, crossCheck

        def match(self, q, t):
            assert self.cross
            rule = orc.crosscheck_mutual_from_matrix if mutual else orc.crosscheck_from_matrix    # not orc.bf_match: bench rebinds it
            return list(orc.to_dmatches(*rule(orc.hamming_matrix(q, t, self.norm))))

        def knnMatch(self, q, t, k):
            idx, dist = orc.knn2_from_matrix(orc.hamming_matrix(q, t, self.norm))
            return [[orc.DMatch(i, idx[i, j], dist[i, j]) for j in range(idx.shape[1])] for i in range(len(q))]

    def solve_pnp(obj, img, calib, dist, flags=0):
        with np.errstate(all="ignore"):
            return (orc.solve_pnp_ap3p if flags == orc.SOLVEPNP_AP3P else orc.solve_pnp_iterative)(obj, img, calib)

    cv2.BFMatcher, cv2.solvePnP = BFMatcher, solve_pnp
    cv2.Rodrigues = lambda r: (orc.rodrigues_vec_to_mat(np.asarray(r).reshape(3)), None)
    return cv2


def _check(cv2, subject, want_rule=None, **kw):
    rep = cv2_pin.report(cv2, subject, **kw)
    rules, findings = cv2_pin.verdict(rep)
    print(json.dumps(rep))
    print("cross-check rule of this cv2:", rules)
    assert not findings, findings
    if want_rule:
        assert set(rules.values()) == {want_rule}
    return rules


def test_harness_against_the_stand_in_with_either_rule():
    _check(_standin(), cv2_pin.OracleSubject(), "ocv_legacy", n_ap3p=40, n_refit=3)
    rep = cv2_pin.report(_standin(mutual=True), cv2_pin.OracleSubject(), n_ap3p=5, n_refit=1)
    assert set(cv2_pin.verdict(rep)[0].values()) == {"mutual"}
    for norm in ("NORM_HAMMING", "NORM_HAMMING2"):                 # and the two rules are told apart by the cases
        assert rep[norm]["cross_check_equal"]["ocv_legacy"] < rep[norm]["cases"]


@pytest.mark.gpu
def test_hip_path_against_the_stand_in():
    _check(_standin(), cv2_pin.HipSubject(), "ocv_legacy", n_ap3p=60, n_refit=4)


def test_oracle_against_the_installed_cv2():
    cv2 = pytest.importorskip("cv2")
    _check(cv2, cv2_pin.OracleSubject())


@pytest.mark.gpu
def test_hip_path_against_the_installed_cv2():
    cv2 = pytest.importorskip("cv2")
    _check(cv2, cv2_pin.HipSubject())


def test_bench_cv2_leg_runs_the_reference_loop_on_the_given_cv2():
    """bench.py's optional `cpu_baseline.cv2_reference_path` (only where cv2 imports): the oracle's per-pair driver with
    BFMatcher / solvePnP / Rodrigues bound to the cv2 module it is given.  With the stand-in the result must be the plain
    oracle's, and the port's own bindings must be back afterwards."""
    import bench
    from slam_amd.synth import SynthConfig, SynthSequence
    cfg = SynthConfig(n_frames=12, n_features=260, seed=3, spawn_per_frame=30, ragged=7)
    desc, xy, count = SynthSequence(cfg).frames(0, 3)

    def frame(i):
        n0, n1 = count[i]
        return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])

    for p in range(2):
        _, it_a, kp_a, mask_a, T_a = bench._oracle_pair((frame(p), frame(p + 1), 1000 + p))
        _, it_b, kp_b, mask_b, T_b = bench._oracle_pair((frame(p), frame(p + 1), 1000 + p), cv2=_standin())
        assert len(kp_a) > 50 and it_a == it_b and np.array_equal(kp_a, kp_b) and np.array_equal(mask_a, mask_b)
        assert np.array_equal(T_a, T_b)
    o = bench._oracle()
    assert o.pnp is o._port_pnp and o.bf_match.__name__ == "<lambda>"
    assert bench._try_cv2() is None or hasattr(bench._try_cv2(), "BFMatcher")

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
            return list(orc.to_dmatches(*(orc.bf_match_mutual if mutual else orc.bf_match)(q, t, self.norm)))

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

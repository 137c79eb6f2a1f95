"""Trajectory evaluation against vectors recorded from the reference's own functions
(tests/golden/make_golden_trajectory.py -> trajectory_vectors.npz).  Host code only: needs the library for nothing
but the package import."""
import os

import numpy as np
import pytest

from slam_amd import trajectory as tj

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "trajectory_vectors.npz"))


def test_series_match_the_reference_bit_for_bit():
    est, gt = G["est"], G["gt"]
    assert np.array_equal(np.array([tj.between(a, b) for a, b in zip(est, gt)]), G["between"])
    assert np.array_equal(tj.location_errors(est, gt), G["loc_err"])
    assert np.array_equal(tj.angle_errors(est, gt), G["ang_err"])
    assert np.array_equal(np.array([tj.invert(t) for t in est]), G["inv"])


def test_euler_angles_incl_singular_branch():
    got = np.array([tj.euler_angles(r) for r in G["euler_R"]])
    assert np.array_equal(got, G["euler"])
    assert (got[-2:, 2] == 0).all()                         # the two gimbal-singular inputs took the z = 0 branch


def test_random_poses_between_and_centres():
    T = G["rand_T"]
    assert np.array_equal(np.array([tj.between(T[i], T[(i + 7) % len(T)]) for i in range(len(T))]), G["rand_between"])
    assert np.array_equal(np.array([tj.centre(t) for t in T]), G["rand_centres"])


def test_utills_wrappers_keep_the_reference_names():
    pytest.importorskip("torch")
    from slam_amd import utills as U
    est = G["est"]
    assert np.array_equal(np.array([U.translation_to_vec(t) for t in est]), G["vec"])
    assert np.array_equal(U.left_cameras_angles(est), np.array([tj.euler_angles(t[:, :3]) for t in est]))
    assert np.array_equal(np.array([U.between(a, b) for a, b in zip(est, G["gt"])]), G["between"])

    class _DB:                                              # get_relative_cam_trans is all the two functions use
        def __init__(self, rel): self.rel = rel
        def get_relative_cam_trans(self): return self.rel
    rel = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))["movie_rel"]
    db = _DB(list(rel))
    assert np.array_equal(U.cameras_rel_pnp_est_and_ground_truth_locations(db, G["gt"]), G["loc_err"])
    assert np.array_equal(U.cameras_rel_pnp_est_and_ground_truth_angles(db, G["gt"]), G["ang_err"])
    with pytest.raises(ValueError):
        U.cameras_rel_pnp_est_and_ground_truth_angles(db)
    s = U.trajectory_error_summary(db, G["gt"])
    assert s["frames"] == len(rel) and s["ate_rmse_m"] < 0.05


def test_summary_properties():
    rng = np.random.default_rng(3)

    def chain(n, noise):
        T, out = np.eye(4), []
        for _ in range(n):
            w = np.array([0.0, 0.02, 0.0]) + rng.normal(0, noise, 3)
            th = np.linalg.norm(w)
            k = w / th
            Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
            S = np.eye(4)
            S[:3, :3], S[:3, 3] = R, np.array([0.0, 0.0, -1.0]) + rng.normal(0, noise, 3)
            T = S @ T
            out.append(T[:3].copy())
        return np.array(out)

    gt = chain(50, 0.0)
    z = tj.summary(gt, gt)
    # between(T, T) is R^T R in floating point, not exactly the identity: zero up to rounding
    assert z["ate_rmse_m"] < 1e-12 and z["rpe_trans_rmse_m"] < 1e-12 and z["rpe_rot_rmse_deg"] < 1e-5
    est = chain(50, 0.002)
    s1, s5 = tj.summary(est, gt, 1), tj.summary(est, gt, 5)
    assert 0 < s1["rpe_trans_rmse_m"] < s5["rpe_trans_rmse_m"] < s1["ate_max_m"] * 5
    assert s1["end_drift_m"] == pytest.approx(np.linalg.norm(tj.centre(tj.between(est[-1], gt[-1]))))
    # RPE ignores a common rigid motion of the past: perturbing only the first pose leaves later segments untouched
    te, re, seg = tj.rpe(est, gt, 1)
    assert seg == pytest.approx(np.linalg.norm(np.diff(np.array([tj.centre(t) for t in gt]), axis=0), axis=1))
    with pytest.raises(ValueError):
        tj.rpe(est, gt, 0)
    with pytest.raises(ValueError):
        tj.rpe(est, gt, 50)

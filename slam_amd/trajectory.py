"""Trajectory evaluation as numbers.

The reference only plots these series (Excercises/project.py:86-135 through utils/utills.py:1200-1222): per frame, the
pose difference `between(estimate, ground truth)` (utills.py:653-656), its camera centre (:599-606) and the Euler
angles of `R_est^T R_gt` (:1072-1086).  This module returns the same series - bit for bit, tests/test_trajectory_cpu.py
holds them to vectors recorded from the reference - and condenses them into ATE / RPE figures so that "the
trajectory agrees with the reference's" can be stated as a number.  Pure host code, float64, no device work.
"""
from __future__ import annotations

import math

import numpy as np


def compose(first: np.ndarray, second: np.ndarray) -> np.ndarray:
    """compose_transformations (utills.py:659-669): second @ [first; 0 0 0 1]."""
    return second @ np.vstack((first, np.array([0, 0, 0, 1])))


def invert(ex_cam: np.ndarray) -> np.ndarray:
    """convert_ex_cam_to_cam_to_world (utills.py:1025-1034)."""
    R, t = ex_cam[:, :3], ex_cam[:, 3]
    return np.hstack((R.T, (-R.T @ t).reshape(3, 1)))


def between(first_cam: np.ndarray, second_cam: np.ndarray) -> np.ndarray:
    """utills.py:653-656: the transformation that takes `first_cam` to `second_cam`."""
    return compose(invert(first_cam), second_cam)


def centre(ex_cam: np.ndarray) -> np.ndarray:
    """relative_camera_pos (utills.py:599-606)."""
    return -1 * ex_cam[:, :3].T @ ex_cam[:, 3]


def euler_angles(R: np.ndarray) -> np.ndarray:
    """rot_mat_to_euler_angles (utills.py:1072-1086), including its singular branch."""
    sy = math.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])
    if not sy < 1e-6:
        x = math.atan2(R[2, 1], R[2, 2])
        y = math.atan2(-R[2, 0], sy)
        z = math.atan2(R[1, 0], R[0, 0])
    else:
        x = math.atan2(-R[1, 2], R[1, 1])
        y = math.atan2(-R[2, 0], sy)
        z = 0
    return np.array([x, y, z])


def location_errors(est_global, gt_global) -> np.ndarray:
    """float64[n, 3]: the series of cameras_rel_pnp_est_and_ground_truth_locations (utills.py:1200-1208)."""
    return np.array([centre(between(a, b)) for a, b in zip(est_global, gt_global)]).reshape(-1, 3)


def angle_errors(est_global, gt_global) -> np.ndarray:
    """float64[n, 3] radians: the series of cameras_rel_pnp_est_and_ground_truth_angles (utills.py:1211-1222)."""
    return np.array([euler_angles(a[:, :3].T @ b[:, :3]) for a, b in zip(est_global, gt_global)]).reshape(-1, 3)


def _rot_angle(R: np.ndarray) -> float:
    return math.acos(min(1.0, max(-1.0, (np.trace(R) - 1.0) / 2.0)))


def rpe(est_global, gt_global, delta: int = 1):
    """Relative pose error over `delta` frames: per i, E = between(between(gt_i, gt_{i+d}), between(est_i, est_{i+d})).
    Returns (translation error [m] float64[n-d], rotation error [rad] float64[n-d], ground-truth path length of the
    segment [m] float64[n-d])."""
    n = min(len(est_global), len(gt_global))
    if not 0 < delta < n:
        raise ValueError("delta must be in [1, n)")
    te, re, seg = np.empty(n - delta), np.empty(n - delta), np.empty(n - delta)
    gc = np.array([centre(t) for t in gt_global[:n]])
    step = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(gc, axis=0), axis=1))])
    for i in range(n - delta):
        E = between(between(gt_global[i], gt_global[i + delta]), between(est_global[i], est_global[i + delta]))
        te[i], re[i] = np.linalg.norm(E[:, 3]), _rot_angle(E[:, :3])
        seg[i] = step[i + delta] - step[i]
    return te, re, seg


def summary(est_global, gt_global, delta: int = 1) -> dict:
    """ATE (camera-centre RMSE, no alignment: both trajectories start at the reference's first camera, as in the
    reference's own comparison), the end-point drift, and RPE over `delta` frames."""
    loc = location_errors(est_global, gt_global)
    d = np.linalg.norm(loc, axis=1)
    te, re, seg = rpe(est_global, gt_global, delta)
    path = float(seg.sum() / delta) if delta else 0.0
    return {
        "frames": int(len(d)),
        "ate_rmse_m": float(np.sqrt(np.mean(d * d))),
        "ate_max_m": float(d.max()),
        "end_drift_m": float(d[-1]),
        "end_drift_pct_of_path": float(100.0 * d[-1] / path) if path > 0 else float("nan"),
        "rpe_delta": int(delta),
        "rpe_trans_rmse_m": float(np.sqrt(np.mean(te * te))),
        "rpe_trans_pct": float(100.0 * te.sum() / seg.sum()) if seg.sum() > 0 else float("nan"),
        "rpe_rot_rmse_deg": float(np.degrees(np.sqrt(np.mean(re * re)))),
        "rpe_rot_deg_per_m": float(np.degrees(re.sum()) / seg.sum()) if seg.sum() > 0 else float("nan"),
    }

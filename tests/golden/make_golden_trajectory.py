"""Generates tests/golden/trajectory_vectors.npz by running the REFERENCE's own trajectory-evaluation functions
(utils/utills.py `between`, `convert_ex_cam_to_cam_to_world`, `left_cameras_trajectory`,
`left_cameras_angles` / `rot_mat_to_euler_angles`, `translation_to_vec`, and the bodies of
`cameras_rel_pnp_est_and_ground_truth_locations` / `_angles`, whose only other ingredient is the KITTI poses file).

Build container only (needs /root/reference), same stand-ins as make_golden.py:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_trajectory.py

Data only: inputs (poses) and the reference's outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs nothing until asked)
import slam_oracle as orc  # noqa: E402
from slam_amd.synth import SynthConfig, SynthSequence  # noqa: E402


def main():
    mg.install_standins({})
    from utils import utills as U          # THE REFERENCE

    g = {}
    z = np.load(os.path.join(HERE, "reference_vectors.npz"))
    est = z["movie_glob"]                                              # the reference's own 8-frame trajectory
    seq = SynthSequence(SynthConfig(n_frames=40, n_features=320, seed=7, spawn_per_frame=40, ragged=13))
    gt = np.array([seq.gt_global(i) for i in range(len(est))])
    g["est"], g["gt"] = est, gt
    diffs = [U.between(a, b) for a, b in zip(est, gt)]
    g["between"] = np.array(diffs)
    g["loc_err"] = U.left_cameras_trajectory(diffs)                     # body of :1200-1208
    g["ang_err"] = U.left_cameras_angles([np.hstack([a[:, :3].T @ b[:, :3], np.zeros((3, 1))]) for a, b in zip(est, gt)])
    g["inv"] = np.array([U.convert_ex_cam_to_cam_to_world(t) for t in est])
    g["vec"] = np.array([U.translation_to_vec(t) for t in est])

    # random poses, large rotations, and the gimbal-singular branch of rot_mat_to_euler_angles (:1075, :1082-1084)
    rng = np.random.default_rng(11)
    R = [orc.rodrigues_vec_to_mat(rng.normal(0, 1.2, 3)) for _ in range(30)]
    for s in (1.0, -1.0):
        c = 1e-9
        R.append(np.array([[c, 0.0, s * 1.0], [0.3 * c, 1.0, 0.0], [-s * 1.0, 0.0, c]]))     # sy < 1e-6
    R = np.array(R)
    g["euler_R"] = R
    g["euler"] = np.array([U.rot_mat_to_euler_angles(r) for r in R])
    T = np.concatenate([R, rng.normal(0, 5, (len(R), 3, 1))], axis=2)
    g["rand_T"] = T
    g["rand_between"] = np.array([U.between(T[i], T[(i + 7) % len(T)]) for i in range(len(T))])
    g["rand_centres"] = U.left_cameras_trajectory(T)
    np.savez_compressed(os.path.join(HERE, "trajectory_vectors.npz"), **g)
    print("wrote", len(g), "arrays; loc_err max", float(np.abs(g["loc_err"]).max()))


if __name__ == "__main__":
    main()

"""Deterministic synthetic KITTI-shaped stereo detections.

The reference reads KITTI odometry sequence 00 and runs cv2 AKAZE on every image
(DataDirectory/KittiData.py:72-97, utils/utills.py:46-54).  Neither the dataset
nor cv2 exists offline, and detection is out of scope for the hot path, so this
module fabricates what detection would hand to the hot path: per image a set of
key-points (float32 pixel coordinates) with 61-byte binary descriptors (AKAZE
MLDB, 486 bits), for a stereo rig with the KITTI seq-00 calibration driving a
smooth trajectory through a persistent cloud of landmarks.

Every frame is a pure function of (config, frame index), so ranks generate their
own frame ranges without communication.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

K_KITTI = np.array([[718.856, 0.0, 607.1928], [0.0, 718.856, 185.2157], [0.0, 0.0, 1.0]])
BASELINE_M = 0.5371657188644179
M1_KITTI = np.hstack([np.eye(3), np.zeros((3, 1))])
M2_KITTI = np.hstack([np.eye(3), np.array([[-BASELINE_M], [0.0], [0.0]])])

DESC_BYTES = 61          # AKAZE MLDB: 486 bits
DESC_PAD = 64            # row stride used on the device


@dataclass(frozen=True)
class SynthConfig:
    n_frames: int = 3450
    n_features: int = 2000        # nominal features per image (ragged: a few less)
    true_frac: float = 0.7        # share of features that are projections of landmarks
    seed: int = 1234
    px_noise: float = 0.3         # detection noise, pixels (1 sigma)
    flip_and: int = 5             # descriptor noise: bit flips with p = 2**-flip_and per bit
    speed: float = 0.8            # metres per frame
    yaw_sigma_deg: float = 0.5
    depth_min: float = 4.0
    depth_max: float = 80.0
    spawn_per_frame: int = 48     # new landmarks created in front of every camera
    window: int = 110             # frames either side whose landmarks are tested for visibility
    ragged: int = 61              # image i holds n_features - (hash % ragged) features
    width: int = 1241
    height: int = 376
    mismatch_frac: float = 0.0    # share of landmark features whose right-image descriptor is swapped (stress)
    gross_frac: float = 0.04      # share of landmark detections per image with gross localisation error
    gross_sigma: float = 6.0      # pixels (1 sigma) of that gross error
    gross_x_only: bool = False    # gross error along the image row only: it survives the stereo-row test (stress)

    @property
    def n_max(self) -> int:
        return (self.n_features + 63) // 64 * 64


def stress_config(n_frames: int = 40, n_features: int = 8000, seed: int = 5) -> SynthConfig:
    """BASELINE configs[3] / SURVEY 8(d) "Stress": 8k features per image, 60 % of the tracked correspondences outliers
    of the pose (gross localisation errors along the image row, 15 px at 1 sigma, in 27 % of the detections: they pass
    the stereo-row test, and a correspondence is an inlier only if all four of its detections are clean), T >= 3000
    tracked correspondences per pair."""
    return SynthConfig(n_frames=n_frames, n_features=n_features, true_frac=0.85, seed=seed, spawn_per_frame=800,
                       mismatch_frac=0.05, gross_frac=0.27, gross_sigma=15.0, gross_x_only=True, ragged=31)


def loop_candidate_pairs(n_frames: int, seed: int = 3):
    """BASELINE configs[4] / SURVEY 8(d) "Config 5": every frame from 6 on against 5 candidate frames, 2 true revisits
    (f - 2, f - 5) + 3 random earlier frames.  Returns (candidate frame, current frame) int32 arrays: the candidate is
    "frame 0" of the pair and the current frame "frame 1" (utils/utills.py:1128-1134)."""
    rng = np.random.default_rng(seed)
    cur, cand = [], []
    for f in range(6, n_frames):
        for c in (f - 2, f - 5, *rng.integers(0, max(1, f - 5), 3)):
            cand.append(int(c))
            cur.append(f)
    return np.array(cand, np.int32), np.array(cur, np.int32)


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


class SynthSequence:
    """Landmark world + trajectory; `frame(i)` renders the detections of stereo frame i."""

    def __init__(self, cfg: SynthConfig):
        self.cfg = cfg
        rng = np.random.default_rng([cfg.seed, 0xA11CE])
        F = cfg.n_frames
        # camera-to-world trajectory: forward along the camera z axis, yaw random walk
        yaw = np.cumsum(rng.normal(0.0, np.deg2rad(cfg.yaw_sigma_deg), F))
        yaw[0] = 0.0
        step = cfg.speed * (1.0 + 0.15 * rng.uniform(-1, 1, F))
        pos = np.zeros((F, 3))
        R_wc = np.empty((F, 3, 3))
        for i in range(F):
            R_wc[i] = _rot_y(yaw[i])
            if i:
                pos[i] = pos[i - 1] + R_wc[i - 1] @ np.array([0.0, 0.0, step[i]])
        pos[:, 1] += 0.02 * np.cumsum(rng.normal(0, 1, F)) * 0.1
        self.R_cw = np.transpose(R_wc, (0, 2, 1))                   # world -> camera
        self.t_cw = -np.einsum("fij,fj->fi", self.R_cw, pos)
        # landmarks: spawned in front of every camera, uniformly over the image and in depth
        S = cfg.spawn_per_frame
        u = rng.uniform(-0.15 * cfg.width, 1.15 * cfg.width, (F, S))
        v = rng.uniform(0, cfg.height, (F, S))
        z = rng.uniform(cfg.depth_min, cfg.depth_max, (F, S))
        xc = np.stack([(u - K_KITTI[0, 2]) / K_KITTI[0, 0] * z, (v - K_KITTI[1, 2]) / K_KITTI[1, 1] * z, z], axis=2)
        self.lm_xyz = np.einsum("fij,fsj->fsi", R_wc, xc) + pos[:, None, :]      # [F,S,3] world
        self.lm_prio = rng.random((F, S))
        base = rng.integers(0, 256, (F, S, DESC_BYTES), dtype=np.uint8)
        base[:, :, DESC_BYTES - 1] &= 0x3F                                      # 486 valid bits
        self.lm_desc = base

    # ground truth ---------------------------------------------------------------------------
    def gt_global(self, i: int) -> np.ndarray:
        """World -> left camera i, [R|t] 3x4, relative to camera 0 (which is [I|0])."""
        return np.hstack([self.R_cw[i], self.t_cw[i][:, None]])

    def gt_relative(self, i: int) -> np.ndarray:
        """Pose of camera i relative to camera i-1 (what the hot path estimates); i >= 1."""
        R = self.R_cw[i] @ self.R_cw[i - 1].T
        t = self.t_cw[i] - R @ self.t_cw[i - 1]
        return np.hstack([R, t[:, None]])

    def n_in_image(self, i: int, side: int) -> int:
        cfg = self.cfg
        h = (i * 2654435761 + side * 40503 + cfg.seed * 97) & 0xFFFFFFFF
        return cfg.n_features - (h % cfg.ragged if cfg.ragged > 1 else 0)

    # rendering ------------------------------------------------------------------------------
    def frame(self, i: int):
        """Returns dict(desc uint8[2,n_max,64], xy float32[2,n_max,2], count int32[2],
        lm int32[2,n_max] landmark id per feature or -1)."""
        cfg = self.cfg
        rng = np.random.default_rng([cfg.seed, 0xF00D, i])
        lo, hi = max(0, i - cfg.window), min(cfg.n_frames, i + cfg.window + 1)
        S = cfg.spawn_per_frame
        xyz = self.lm_xyz[lo:hi].reshape(-1, 3)
        ids = (np.arange(lo, hi)[:, None] * S + np.arange(S)[None, :]).reshape(-1)
        prio = self.lm_prio[lo:hi].reshape(-1)
        xc = xyz @ self.R_cw[i].T + self.t_cw[i]
        fx, fy, cx, cy = K_KITTI[0, 0], K_KITTI[1, 1], K_KITTI[0, 2], K_KITTI[1, 2]
        zc = np.where(xc[:, 2] > 1e-6, xc[:, 2], 1.0)
        ul = fx * xc[:, 0] / zc + cx
        ur = fx * (xc[:, 0] - BASELINE_M) / zc + cx
        vv = fy * xc[:, 1] / zc + cy
        vis = (xc[:, 2] > 2.0) & (xc[:, 2] < 1.25 * cfg.depth_max) & (ul >= 1) & (ul < cfg.width - 1) & \
              (ur >= 1) & (ur < cfg.width - 1) & (vv >= 1) & (vv < cfg.height - 1)
        cand = np.nonzero(vis)[0]
        n_img = [self.n_in_image(i, 0), self.n_in_image(i, 1)]
        n_true = min(len(cand), int(round(cfg.true_frac * min(n_img))))
        if len(cand) > n_true:
            cand = cand[np.argsort(-prio[cand], kind="stable")[:n_true]]
        lm_ids = ids[cand]
        base = self.lm_desc.reshape(-1, DESC_BYTES)[lm_ids]

        desc = np.zeros((2, cfg.n_max, DESC_PAD), np.uint8)
        xy = np.zeros((2, cfg.n_max, 2), np.float32)
        lm = np.full((2, cfg.n_max), -1, np.int32)
        for side, uu in ((0, ul), (1, ur)):
            n = n_img[side]
            d = np.empty((n, DESC_BYTES), np.uint8)
            p = np.empty((n, 2), np.float64)
            noise = rng.integers(0, 256, (cfg.flip_and, n_true, DESC_BYTES), dtype=np.uint8)
            flips = np.bitwise_and.reduce(noise, axis=0)
            flips[:, DESC_BYTES - 1] &= 0x3F
            d[:n_true] = base ^ flips
            if side == 1 and cfg.mismatch_frac > 0 and n_true > 1:
                k = int(cfg.mismatch_frac * n_true)
                sel = rng.choice(n_true, k, replace=False)
                d[sel] = d[np.roll(sel, 1)]
            p[:n_true, 0] = uu[cand] + rng.normal(0, cfg.px_noise, n_true)
            p[:n_true, 1] = vv[cand] + rng.normal(0, cfg.px_noise, n_true)
            if cfg.gross_frac > 0 and n_true:
                g = rng.random(n_true) < cfg.gross_frac
                e = rng.normal(0, cfg.gross_sigma, (int(g.sum()), 2))
                if cfg.gross_x_only:
                    e[:, 1] = 0.0
                p[:n_true][g] += e
            n_cl = n - n_true
            cl = rng.integers(0, 256, (n_cl, DESC_BYTES), dtype=np.uint8)
            cl[:, DESC_BYTES - 1] &= 0x3F
            d[n_true:] = cl
            p[n_true:, 0] = rng.uniform(0, cfg.width, n_cl)
            p[n_true:, 1] = rng.uniform(0, cfg.height, n_cl)
            ident = np.full(n, -1, np.int32)
            ident[:n_true] = lm_ids
            perm = rng.permutation(n)
            desc[side, :n, :DESC_BYTES] = d[perm]
            xy[side, :n] = p[perm].astype(np.float32)
            lm[side, :n] = ident[perm]
        return dict(desc=desc, xy=xy, count=np.array(n_img, np.int32), lm=lm)

    def frames(self, start: int, stop: int):
        """Stacked detections of frames [start, stop): desc[F,2,n_max,64], xy[F,2,n_max,2], count[F,2]."""
        n = stop - start
        cfg = self.cfg
        desc = np.zeros((n, 2, cfg.n_max, DESC_PAD), np.uint8)
        xy = np.zeros((n, 2, cfg.n_max, 2), np.float32)
        count = np.zeros((n, 2), np.int32)
        for k in range(n):
            f = self.frame(start + k)
            desc[k], xy[k], count[k] = f["desc"], f["xy"], f["count"]
        return desc, xy, count

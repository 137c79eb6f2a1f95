"""Batched shard pipeline: all frame pairs of a frame range through libslam_amd's slam_track_pairs.

This is the host-side counterpart of the loop in find_features_in_consecutive_frames_whole_movie
(utils/utills.py:834-873): detections in (device tensors), relative poses + supporter sets out (numpy)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, ops
from .config import HotPathConfig


@dataclass
class ShardResult:
    pose: np.ndarray          # float64[n_pairs, 3, 4]; NaN where the reference returns None
    stats: np.ndarray         # int32[n_pairs, 8]: T, supporters, iterations, draws, best draw, status, n stereo0, n temporal
    mask: np.ndarray          # bool[n_pairs, n_max]  supporters among the tracked correspondences (pre-refit)
    tracked_kp: torch.Tensor  # int32[n_pairs, n_max, 4] on the device: key-point indices (L0, R0, L1, R1)
    timing_ms: np.ndarray     # float64[8], see slam_track_last_timing

    @property
    def tracked(self):
        return self.stats[:, 0]

    @property
    def supporters(self):
        return self.stats[:, 1]

    @property
    def ok(self):
        return (self.stats[:, 5] & _lib.SLAM_PAIR_OK) != 0

    def inliers_percentage(self):
        """utills.py:923."""
        t = np.maximum(self.tracked, 1)
        return 100.0 * self.supporters / t


def _track_config(cfg: HotPathConfig) -> _lib.SlamTrackConfig:
    c = _lib.SlamTrackConfig()
    c.K[:] = np.asarray(cfg.K, np.float64).reshape(9).tolist()
    c.M1[:] = np.asarray(cfg.M1, np.float64).reshape(12).tolist()
    c.M2[:] = np.asarray(cfg.M2, np.float64).reshape(12).tolist()
    c.rec_err, c.acc, c.prob, c.max_iterations = cfg.rec_err, cfg.supp_err, cfg.prob, cfg.max_iterations
    c.fixed_iterations, c.max_draws, c.norm, c.first_chunk = cfg.fixed_iterations, cfg.max_draws, cfg.norm, cfg.first_chunk
    c.match_mode, c.ratio = cfg.match_mode(), float(cfg.ratio)
    return c


class ShardTracker:
    """Owns the device/pinned workspaces for shards of up to (n_frames, n_pairs, n_max) and runs them.

    n_chunks = 0 (default) leaves the plan to the library (two blocks for a large shard, one for a small one);
    n_chunks > 1 asks the library to split the pair list into contiguous blocks and pipeline them over two internal
    HIP streams, so the latency-bound RANSAC rounds (small kernels, host replay, D2H syncs) of one block overlap the
    VALU-bound matching of the next.  Results are identical to n_chunks = 1: every pair's computation is independent
    of which block carries it."""

    MAX_PAIRS_PER_CALL = 65535      # slam_track_pairs' limit; run() hands longer pair lists over in blocks of this size

    def __init__(self, n_frames: int, n_pairs: int, n_max: int, cfg: HotPathConfig | None = None, n_chunks: int = 0,
                 taper: float = 1.0, serial_match: bool | None = None):
        self.cfg = cfg or HotPathConfig()
        self.n_frames, self.n_pairs, self.n_max = n_frames, n_pairs, n_max
        if n_max % 64:
            raise ValueError("n_max must be a multiple of 64")
        self.dev = ops.device()
        self._c = _track_config(self.cfg)
        self._c.pipeline_chunks = max(0, int(n_chunks))                # 0: the library's own plan for the shard size
        self._c.pipeline_taper_permille = int(round(1000 * taper))     # each chunk = taper x the previous one
        # chunks' matchers back to back instead of side by side (None: the default plan)
        self._c.pipeline_serial_match = int(bool(serial_match)) if serial_match is not None else 0
        L = _lib.lib()
        per_call = min(n_pairs, self.MAX_PAIRS_PER_CALL)          # the workspaces serve one call (one block) at a time
        self.d_ws_bytes = L.slam_track_device_workspace_bytes(n_frames, per_call, n_max, ctypes.byref(self._c))
        self.h_ws_bytes = L.slam_track_host_workspace_bytes(n_frames, per_call, n_max, ctypes.byref(self._c))
        self.d_ws = torch.empty(self.d_ws_bytes, dtype=torch.uint8, device=self.dev)
        self.h_ws = torch.empty(self.h_ws_bytes, dtype=torch.uint8).pin_memory()
        self.tracked_kp = torch.zeros((n_pairs, n_max, 4), dtype=torch.int32, device=self.dev)
        self.h_pose = torch.empty((n_pairs, 12), dtype=torch.float64).pin_memory()
        self.h_stats = torch.empty((n_pairs, 8), dtype=torch.int32).pin_memory()
        self.h_mask = torch.empty((n_pairs, n_max // 64), dtype=torch.int64).pin_memory()

    def run(self, desc: torch.Tensor, xy: torch.Tensor, count: torch.Tensor, pair_f0: np.ndarray, pair_f1: np.ndarray,
            seeds: np.ndarray, unpack_mask: bool = True, stereo_key: torch.Tensor | None = None,
            copy: bool = True) -> ShardResult:
        """desc uint8[2F, n_max, 64], xy float32[2F, n_max, 2], count int32[2F] on the device (image 2f = left of
        frame f, 2f+1 = right); pair_f0/f1 int32[P] frame indices; seeds uint32[P] (np.random.seed per pair).
        stereo_key (optional) int32[F, n_max] on the device: the frames' stereo keys computed earlier (`stereo_keys`,
        possibly on another rank); the call then matches only the temporal problems and reads no right-image descriptor.
        copy=False: the result's host arrays are views of the tracker's page-locked output buffers (where the library wrote
        them), valid until this tracker's next run -- for callers that consume a shard's results before they start the next."""
        F = desc.shape[0] // 2
        P = len(pair_f0)
        if F > self.n_frames or P > self.n_pairs or desc.shape[1] != self.n_max:
            raise ValueError("shard larger than the tracker's workspace")
        if stereo_key is not None:
            if tuple(stereo_key.shape) != (F, self.n_max):
                raise ValueError("stereo_key must hold one row of n_max keys per frame")
            ops._dev(stereo_key, torch.int32)
        f0 = np.ascontiguousarray(pair_f0, dtype=np.int32)
        f1 = np.ascontiguousarray(pair_f1, dtype=np.int32)
        sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64) & 0xFFFFFFFF, dtype=np.uint32)
        d_desc, d_xy, d_count = desc, xy, count
        ops._dev(d_desc, torch.uint8), ops._dev(d_xy, torch.float32), ops._dev(d_count, torch.int32)     # device / dtype / layout checks
        L = _lib.lib()
        tm = np.zeros(8)
        M = self.MAX_PAIRS_PER_CALL
        for a in range(0, max(P, 1), M):
            e = min(a + M, P)
            # one call when the list fits (every frame handed over, as before); otherwise block by block, each with
            # the frame range its pairs touch, indices rebased to it
            lo, hi = (0, F) if P <= M else (int(min(f0[a:e].min(), f1[a:e].min())), int(max(f0[a:e].max(), f1[a:e].max())) + 1)
            f0b, f1b = np.ascontiguousarray(f0[a:e] - lo), np.ascontiguousarray(f1[a:e] - lo)
            sdb = np.ascontiguousarray(sd[a:e])
            io = _lib.SlamTrackIO()
            io.d_desc = d_desc.data_ptr() + 2 * lo * self.n_max * 64
            io.d_xy = d_xy.data_ptr() + 2 * lo * self.n_max * 8
            io.d_count = d_count.data_ptr() + 2 * lo * 4
            io.n_frames, io.n_max = hi - lo, self.n_max
            io.h_pair_f0, io.h_pair_f1, io.h_pair_seed = f0b.ctypes.data, f1b.ctypes.data, sdb.ctypes.data
            io.n_pairs = e - a
            io.d_tracked_kp = self.tracked_kp.data_ptr() + a * self.n_max * 16
            io.h_pose = self.h_pose.data_ptr() + a * 96
            io.h_stats = self.h_stats.data_ptr() + a * 32
            io.h_mask = self.h_mask.data_ptr() + a * (self.n_max // 64) * 8
            io.d_workspace, io.d_workspace_bytes = self.d_ws.data_ptr(), self.d_ws_bytes
            io.h_workspace, io.h_workspace_bytes = self.h_ws.data_ptr(), self.h_ws_bytes
            if stereo_key is None:
                _lib.check(L.slam_track_pairs(ctypes.byref(self._c), ctypes.byref(io), ops._stream()))
            else:
                _lib.check(L.slam_track_pairs_stereo_keys(ctypes.byref(self._c), ctypes.byref(io),
                                                          stereo_key.data_ptr() + lo * self.n_max * 4, ops._stream()))
            t1 = np.zeros(8)
            L.slam_track_last_timing(t1.ctypes.data)
            tm += t1
        pose = self.h_pose[:P].numpy().reshape(P, 3, 4)
        stats = self.h_stats[:P].numpy()
        if unpack_mask:
            mask = ops.mask_words_to_bool(self.h_mask[:P].numpy(), self.n_max)
        else:
            mask = self.h_mask[:P].numpy()
        if copy:
            pose, stats, mask = pose.copy(), stats.copy(), (mask if unpack_mask else mask.copy())
        return ShardResult(pose=pose, stats=stats, mask=mask, tracked_kp=self.tracked_kp[:P], timing_ms=tm)


def stereo_keys(desc: torch.Tensor, count: torch.Tensor, cfg: HotPathConfig | None = None) -> torch.Tensor:
    """The stereo match of every frame on its own (read_and_rec_match's match() call, utils/utills.py:386-396), as the
    packed keys slam_track_pairs computes for itself: desc uint8[2F, n_max, 64], count int32[2F] (image 2f = left of frame
    f, 2f+1 = right) -> int32[F, n_max] on the device.  Feeding them back through ShardTracker.run(stereo_key=...) gives
    the results of the call that matches them itself, bit for bit; a multi-GPU loop-closure pass uses it to match every
    frame on ONE rank (slam_amd.dist.stereo_keys_sharded)."""
    cfg = cfg or HotPathConfig()
    F = desc.shape[0] // 2
    q = torch.arange(0, 2 * F, 2, dtype=torch.int32, device=desc.device)
    mode = cfg.match_mode()
    ratio = (float(cfg.ratio) if cfg.ratio > 0 else 0.6) if mode == _lib.SLAM_MATCH_KNN2 else 0.0
    key1, _ = ops.match_keys_batched(desc, count, q, q + 1, cfg.norm, mode, ratio)
    return key1


def consecutive_pairs(n_frames: int):
    f0 = np.arange(n_frames - 1, dtype=np.int32)
    return f0, f0 + 1


def compose_global(rel: np.ndarray) -> np.ndarray:
    """convert_trans_from_rel_to_global (utils/utills.py:749-757): running left-composition
    T_i @ [T_{i-1..}; 0 0 0 1].  rel float64[n, 3, 4] (rel[0] = the first camera, [I|0])."""
    rel = np.ascontiguousarray(rel, dtype=np.float64)
    out = np.empty_like(rel)
    _lib.check(_lib.lib().slam_compose_global(rel.ctypes.data, len(rel), out.ctypes.data))
    return out


def camera_centres(glob: np.ndarray) -> np.ndarray:
    """left_cameras_trajectory / relative_camera_pos (utils/utills.py:599-606, :785-794): -R^T t."""
    return np.array([-1 * t[:, :3].T @ t[:, 3] for t in glob]).reshape(-1, 3)

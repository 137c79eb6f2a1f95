"""Frame-range sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
on ROCm, "gloo" in the CPU tests).

Consecutive-frame relative poses are independent given the detections (README.md:58 of the reference remarks the
loop "can be done in parallel"; the loop itself is utils/utills.py:850), so rank r owns a contiguous block of frame
pairs and the frames they touch (one halo frame, whose stereo match is recomputed rather than communicated).  There
is no data-path collective; the only exchange is one all-gather of the per-pair results (3x4 relative poses and the
per-pair counters, ~130 B per pair) before rank 0 composes the global trajectory on the host
(convert_trans_from_rel_to_global, utils/utills.py:749-757)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def block_range(n_units: int, rank: int, world: int):
    """Contiguous block partition: rank r owns [r*ceil(n/world), min(n, (r+1)*ceil(n/world)))."""
    per = -(-n_units // world)
    a = min(n_units, rank * per)
    return a, min(n_units, a + per)


def shard_frames(n_frames: int, rank: int, world: int):
    """Pairs [a, b) of a movie with n_frames frames and the frames [a, b] they need.
    Returns (pair_start, pair_stop, frame_start, frame_stop_exclusive)."""
    a, b = block_range(n_frames - 1, rank, world)
    return a, b, a, (b + 1 if b > a else a)


def shard_list(n_units: int, rank: int, world: int):
    """Block partition of a flat work list (loop-closure candidate pairs)."""
    return block_range(n_units, rank, world)


def all_gather_rows(local: torch.Tensor, n_total: int, world: int, rank: int) -> torch.Tensor:
    """Gathers row blocks laid out by `block_range` into one [n_total, ...] tensor on every rank."""
    if world == 1 and not dist.is_initialized():
        return local
    per = -(-n_total // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    return out[:n_total]


_gather_buffers = {}


def _gather_ctx(per: int, world: int, dev):
    """Staging for gather_shard_results, allocated once per (block size, world, device): a page-locked row block each way
    and the two device buffers of the collective, so a step pays for the copies and the all-gather, not for allocations."""
    key = (per, world, str(dev))
    ctx = _gather_buffers.get(key)
    if ctx is None:
        pin = dev.type == "cuda"
        h_in = torch.zeros((per, 128), dtype=torch.uint8)
        h_out = torch.empty((world * per, 128), dtype=torch.uint8)
        ctx = _gather_buffers[key] = (h_in.pin_memory() if pin else h_in, h_out.pin_memory() if pin else h_out,
                                      torch.zeros((per, 128), dtype=torch.uint8, device=dev),
                                      torch.empty((world * per, 128), dtype=torch.uint8, device=dev))
    return ctx


def gather_shard_results(pose_local: np.ndarray, stats_local: np.ndarray, n_pairs_total: int, device=None):
    """pose_local float64[p, 3, 4], stats_local int32[p, 8] of this rank's pair block -> the full arrays on every rank.
    One collective: a pair's pose (96 bytes) and counters (32 bytes) travel as one 128-byte row."""
    if not dist.is_initialized():
        return pose_local, stats_local
    world = dist.get_world_size()
    dev = _coll_device(device)
    p = len(pose_local)
    per = -(-n_pairs_total // world)
    if p > per:
        raise ValueError("more local pairs than a block of the partition holds")
    h_in, h_out, d_in, d_out = _gather_ctx(per, world, dev)
    row = h_in.numpy()
    row[:p, :96] = np.ascontiguousarray(pose_local, dtype=np.float64).reshape(p, 12).view(np.uint8)
    row[:p, 96:] = np.ascontiguousarray(stats_local, dtype=np.int32).reshape(p, 8).view(np.uint8)
    row[p:] = 0
    if dev.type == "cuda":
        d_in.copy_(h_in, non_blocking=True)
        dist.all_gather_into_tensor(d_out, d_in)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        full = h_out.numpy()[:n_pairs_total]
    else:
        dist.all_gather_into_tensor(h_out, h_in)
        full = h_out.numpy()[:n_pairs_total]
    pose = np.ascontiguousarray(full[:, :96]).view(np.float64).reshape(-1, 3, 4)
    stats = np.ascontiguousarray(full[:, 96:]).view(np.int32).reshape(-1, 8)
    return pose, stats


def _coll_device(device=None):
    if device is not None:
        return device
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_ragged_rows(local_rows: np.ndarray, local_counts: np.ndarray, n_units_total: int, device=None):
    """Variable-length row lists per unit (the supporters' key-point rows of every frame pair: what the track database
    on the other side of the path is built from, DataBase.py:39-51).  local_rows [m, ...]: this rank's rows, unit after
    unit; local_counts int[p]: rows per unit of this rank's block (`block_range`).  Returns (rows of all units, unit
    after unit, int64 row pointer [n_units_total + 1]) on every rank.  Two collectives: the counts, then the rows padded
    to the largest block."""
    if not dist.is_initialized():
        ptr = np.zeros(len(local_counts) + 1, np.int64)
        np.cumsum(local_counts, out=ptr[1:])
        return local_rows, ptr
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _coll_device(device)
    counts = all_gather_rows(torch.from_numpy(np.ascontiguousarray(local_counts, dtype=np.int64).reshape(-1, 1)).to(dev),
                             n_units_total, world, rank).cpu().numpy().reshape(-1)
    ptr = np.zeros(n_units_total + 1, np.int64)
    np.cumsum(counts, out=ptr[1:])
    blocks = [block_range(n_units_total, r, world) for r in range(world)]
    rows_of = [int(ptr[b] - ptr[a]) for a, b in blocks]
    m_max = max(rows_of + [1])
    local_rows = np.ascontiguousarray(local_rows)
    if len(local_rows) != rows_of[rank]:
        raise ValueError("local_rows does not hold sum(local_counts) rows")
    pad = torch.zeros((m_max,) + local_rows.shape[1:], dtype=torch.from_numpy(local_rows[:0]).dtype, device=dev)
    pad[:len(local_rows)] = torch.from_numpy(local_rows).to(dev)
    out = torch.empty((world * m_max,) + tuple(pad.shape[1:]), dtype=pad.dtype, device=dev)
    dist.all_gather_into_tensor(out, pad)
    out = out.cpu().numpy()
    return np.concatenate([out[r * m_max:r * m_max + rows_of[r]] for r in range(world)]), ptr


def gather_frames(local_frames: np.ndarray, n_frames: int, device=None) -> np.ndarray:
    """Per-frame arrays of a movie whose frames are held as `shard_frames` hands them out (rank r: the first frames of
    its pairs plus one halo frame).  local_frames [fb - fa, ...] -> [n_frames, ...] on every rank: every rank
    contributes the first frame of each of its pairs; the movie's last frame comes from the rank that owns the last
    pair (its halo)."""
    if not dist.is_initialized():
        return local_frames
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _coll_device(device)
    a, b, _, _ = shard_frames(n_frames, rank, world)
    mine = torch.from_numpy(np.ascontiguousarray(local_frames[:b - a])).to(dev)
    body = all_gather_rows(mine, n_frames - 1, world, rank)
    owner = max(r for r in range(world) if block_range(n_frames - 1, r, world)[1] > block_range(n_frames - 1, r, world)[0]) if n_frames > 1 else 0
    last = (torch.from_numpy(np.ascontiguousarray(local_frames[b - a:b - a + 1])).to(dev) if rank == owner
            else torch.empty((1,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=dev))
    dist.broadcast(last, src=owner)
    return torch.cat([body, last]).cpu().numpy()


def frames_of_block(pair_f0: np.ndarray, pair_f1: np.ndarray, a: int, b: int):
    """The frames a block [a, b) of a flat (frame 0, frame 1) pair list touches, ascending, and the block's two index
    arrays rebased to positions in that list: what a rank reads, stereo-matches and tracks for its block."""
    f0 = np.asarray(pair_f0[a:b], dtype=np.int64)
    f1 = np.asarray(pair_f1[a:b], dtype=np.int64)
    used = np.unique(np.concatenate([f0, f1]))
    return used, np.searchsorted(used, f0).astype(np.int32), np.searchsorted(used, f1).astype(np.int32)


def pair_list_plan(pair_f0, pair_f1, n_frames: int, world: int):
    """How a flat pair list over `n_frames` frames is best split over `world` ranks, in matching problems of the busiest
    rank: one phase (`track_pairs_sharded` without exchange: a rank stereo-matches every frame its block touches, so a
    frame is matched by every rank that needs it) against two phases (every frame stereo-matched on ONE rank, the keys
    all-gathered, then the pair blocks).  -> (two_phase: bool, problems one-phase, problems two-phase).  A pure function
    of the list, so every rank takes the same decision without talking."""
    pair_f0, pair_f1 = np.asarray(pair_f0), np.asarray(pair_f1)
    P = len(pair_f0)
    one = two = 0
    for r in range(world):
        a, b = shard_list(P, r, world)
        fa, fb = block_range(n_frames, r, world)
        used = len(np.unique(np.concatenate([pair_f0[a:b], pair_f1[a:b]]))) if b > a else 0
        one = max(one, used + (b - a))
        two = max(two, (fb - fa) + (b - a))
    return two < 0.9 * one, one, two


def stereo_keys_sharded(n_frames: int, match_block, device=None) -> torch.Tensor:
    """Phase 1 of the two-phase pair-list split -- the one real exchange step of this path (RCCL all-gather over xGMI
    under the nccl backend).  The caller's frame list is block-partitioned (`block_range`); rank r stereo-matches its
    frames, `match_block(frame_start, frame_stop) -> int32[frames, n_max]` packed keys (slam_amd.pipeline.stereo_keys),
    and ONE all-gather leaves int32[n_frames, n_max] on every rank (on the collective's device: the GPU under nccl),
    n_max being the widest block's (narrower blocks are padded with SLAM_KEY_NONE, which is what rows past an image's
    count hold anyway).  28 MB for 3450 frames x 2048 rows."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    fa, fb = block_range(n_frames, rank, world)
    dev = _coll_device(device) if dist.is_initialized() else None
    local = match_block(fa, fb) if fb > fa else torch.empty((0, 64), dtype=torch.int32)
    if local.dtype != torch.int32 or local.dim() != 2 or local.shape[0] != fb - fa:
        raise ValueError("match_block must return int32[frames of the block, n_max]")
    if not dist.is_initialized():
        return local
    n_max = max_over_ranks(local.shape[1], device=device)
    local = local.to(dev)
    if local.shape[1] != n_max:
        wide = torch.full((local.shape[0], n_max), -1, dtype=torch.int32, device=dev)      # SLAM_KEY_NONE
        wide[:, :local.shape[1]] = local
        local = wide
    return all_gather_rows(local, n_frames, world, rank)


def track_pairs_sharded(pair_f0, pair_f1, seeds, run_block, row_width: int, device=None, n_frames: int | None = None,
                        stereo_block=None, plan: str = "auto"):
    """A flat list of independent frame pairs -- the (candidate, current) pairs of loop-closure validation,
    PoseGraph.py:463-472 -> :386-410 -> utils/utills.py:1118-1146 -- block-partitioned over the ranks (`shard_list`).

    Rank r tracks only its block and the frames it touches:
    `run_block(used_frames, f0_local, f1_local, seeds_block, stereo_key) -> (pose[p,3,4], stats[p,8], rows
    int32[m,row_width], rows_per_pair int[p])`, with pair_f0 / pair_f1 the positions in the caller's frame list,
    `used_frames` the ascending positions the block needs and `stereo_key` None (one phase: the block matches its frames'
    stereo pairs itself) or those frames' rows of the all-gathered key table (two phases).  `seeds` (one per pair) are
    rank 0's: every other rank's argument is ignored, so the result is what one process returns for rank 0's random
    stream.

    plan: "one" = no data-path collective (a frame is stereo-matched by every rank whose block touches it); "two" =
    `stereo_keys_sharded(n_frames, stereo_block)` first, so every frame is matched once in the whole job, at the price of
    one all-gather; "auto" = `pair_list_plan` (two phases when they take a tenth or more off the busiest rank's matching
    problems -- candidates spread over the whole past; one when the blocks' frames hardly overlap).  Two phases need
    `n_frames` and `stereo_block`.  Afterwards one all-gather of the 128-byte pose + counter rows and one (ragged) of the
    supporters' rows leave (pose [P,3,4], stats [P,8], rows [M,row_width], row pointer int64[P+1]) on every rank."""
    pair_f0, pair_f1 = np.asarray(pair_f0, np.int32), np.asarray(pair_f1, np.int32)
    P = len(pair_f0)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if plan not in ("auto", "one", "two"):
        raise ValueError("plan must be 'auto', 'one' or 'two'")
    two = plan == "two"
    if plan == "auto" and stereo_block is not None and n_frames is not None and world > 1:
        two = pair_list_plan(pair_f0, pair_f1, n_frames, world)[0]
    if two and (stereo_block is None or n_frames is None):
        raise ValueError("the two-phase plan needs n_frames and stereo_block")
    seeds = broadcast_array(seeds if rank == 0 else None, (P,), np.int64, device=device).astype(np.uint64)
    keys = stereo_keys_sharded(n_frames, stereo_block, device=device) if two else None
    a, b = shard_list(P, rank, world)
    pose_l, stats_l = np.zeros((0, 3, 4)), np.zeros((0, 8), np.int32)
    rows_l, cnt_l = np.zeros((0, row_width), np.int32), np.zeros(0, np.int64)
    if b > a:
        used, l0, l1 = frames_of_block(pair_f0, pair_f1, a, b)
        sk = keys[torch.from_numpy(used).to(keys.device)] if keys is not None else None
        pose_l, stats_l, rows_l, cnt_l = run_block(used, l0, l1, seeds[a:b], sk)
        if len(pose_l) != b - a or len(cnt_l) != b - a or rows_l.shape[1:] != (row_width,):
            raise ValueError("run_block returned arrays of the wrong shape for its block")
    pose, stats = gather_shard_results(pose_l, stats_l, P, device=device)
    rows, ptr = gather_ragged_rows(np.ascontiguousarray(rows_l, dtype=np.int32), np.asarray(cnt_l, np.int64), P, device=device)
    return pose, stats, rows, ptr


def max_over_ranks(value: int, device=None) -> int:
    if not dist.is_initialized():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def broadcast_array(a, shape, dtype, src=0, device=None) -> np.ndarray:
    """One host array from rank `src` to every rank (the per-pair seeds drawn from rank 0's global random stream)."""
    if not dist.is_initialized():
        return np.asarray(a, dtype=dtype).reshape(shape)
    dev = _coll_device(device)
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype).reshape(shape))).to(dev)
    else:
        t = torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def run_sharded(n_frames: int, run_shard, first_pose=None):
    """Runs `run_shard(frame_start, frame_stop, pair_offset) -> (pose[p,3,4], stats[p,8])` on this rank's frame range,
    gathers, and returns (relative poses [n_frames,3,4] incl. the first camera, global poses, stats) on every rank."""
    from .pipeline import compose_global
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    a, b, fa, fb = shard_frames(n_frames, rank, world)
    if b > a:
        pose, stats = run_shard(fa, fb, a)
    else:
        pose, stats = np.zeros((0, 3, 4)), np.zeros((0, 8), np.int32)
    pose, stats = gather_shard_results(pose, stats, n_frames - 1)
    first = np.eye(3, 4) if first_pose is None else np.asarray(first_pose, dtype=np.float64)
    rel = np.concatenate([first[None], pose])
    return rel, compose_global(rel), stats

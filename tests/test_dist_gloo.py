"""world_size-2 gloo test of the frame-range sharding + gather (CPU; the per-shard compute is a stand-in function of
the frame indices, since the HIP path needs a GPU)."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from slam_amd import dist as sdist
from slam_amd.pipeline import compose_global


def _fake_pose(i):
    a = 0.01 * np.sin(i)
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    return np.hstack([R, [[0.001 * i], [0.0], [-0.8 - 0.01 * (i % 7)]]])


def _run_shard(fa, fb, pair_offset):
    n = fb - fa - 1
    pose = np.array([_fake_pose(pair_offset + k) for k in range(n)]).reshape(n, 3, 4)
    stats = np.zeros((n, 8), np.int32)
    stats[:, 0] = np.arange(pair_offset, pair_offset + n)
    stats[:, 1] = fa
    return pose, stats


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rel, glob, stats = sdist.run_sharded(n_frames, _run_shard)
    q.put((rank, rel, glob, stats))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_block_ranges_cover_everything():
    for n in (0, 1, 7, 3449):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                a, b = sdist.block_range(n, r, world)
                cover += list(range(a, b))
            assert cover == list(range(n))
    assert sdist.shard_frames(3450, 7, 8) == (3024, 3449, 3024, 3450)
    assert sdist.shard_frames(3450, 0, 8) == (0, 432, 0, 433)          # one halo frame


def test_two_rank_gather_equals_single_process():
    n_frames = 38                                                     # odd pair count: ragged last block
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_pose, want_stats = _run_shard(0, n_frames, 0)
    want_rel = np.concatenate([np.eye(3, 4)[None], want_pose])
    for rank, rel, glob, stats in got:
        assert np.array_equal(rel, want_rel)
        assert np.array_equal(glob, compose_global(want_rel))
        assert np.array_equal(stats[:, 0], want_stats[:, 0])
        assert set(stats[:, 1]) == {0, 19}                            # which rank's frame range produced each pair


def _ragged_worker(rank, world, port, n_units, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, counts, frames = _ragged_case(n_units, n_frames)
    ptr = np.concatenate([[0], np.cumsum(counts)])
    a, b = sdist.block_range(n_units, rank, world)
    got_rows, got_ptr = sdist.gather_ragged_rows(rows[ptr[a]:ptr[b]], counts[a:b], n_units)
    pa, pb, fa, fb = sdist.shard_frames(n_frames, rank, world)
    got_frames = sdist.gather_frames(frames[fa:fb], n_frames)
    seeds = sdist.broadcast_array(np.arange(5) + 7 if rank == 0 else None, (5,), np.int64)
    widest = sdist.max_over_ranks(64 * (rank + 1))
    q.put((rank, got_rows, got_ptr, got_frames, seeds, widest))
    dist.barrier()
    dist.destroy_process_group()


def _ragged_case(n_units, n_frames):
    rng = np.random.default_rng(4)
    counts = rng.integers(0, 9, n_units).astype(np.int64)
    counts[min(3, n_units - 1)] = 0
    rows = rng.integers(0, 2000, (int(counts.sum()), 4)).astype(np.int32)
    frames = rng.random((n_frames, 2, 8, 2)).astype(np.float32)
    return rows, counts, frames


def test_ragged_rows_and_frames_gather_over_ranks():
    """The exchanges of the sharded whole-movie driver (utills._whole_movie_sharded): supporters' key-point rows of
    every pair (variable length, a pair without any, more ranks than pairs), the frames' coordinates (halo frames, the
    movie's last frame), the seeds from rank 0, the common image width."""
    for world, n_frames in ((2, 12), (3, 6), (3, 3)):
        n_units = n_frames - 1
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, n_units, n_frames, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        rows, counts, frames = _ragged_case(n_units, n_frames)
        for rank, g_rows, g_ptr, g_frames, seeds, widest in got:
            assert g_rows.dtype == np.int32 and np.array_equal(g_rows, rows)
            assert np.array_equal(g_ptr, np.concatenate([[0], np.cumsum(counts)]))
            assert g_frames.dtype == np.float32 and np.array_equal(g_frames, frames)
            assert list(seeds) == [7, 8, 9, 10, 11] and widest == 64 * world

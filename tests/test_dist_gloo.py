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

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


def _worker_digest(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rel, glob, stats = sdist.run_sharded(n_frames, _run_shard)
    q.put((rank, rel.shape, float(np.abs(rel).sum()), float(np.abs(glob).sum()), stats[:, 0].copy(), np.unique(stats[:, 1])))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_on_the_bench_sequence_shape():
    """BASELINE configs[2] at N = 8, the shape the driver's scaling run has: 3450 frames = 3449 pairs in blocks of 432
    (the last rank 425), one halo frame per rank, one all-gather of the 128-byte rows; every rank ends with the whole
    sequence's poses in pair order and the same composed trajectory (stand-in per-shard compute, gloo on the CPU)."""
    n_frames, world = 3450, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_digest, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want_pose, want_stats = _run_shard(0, n_frames, 0)
    want_rel = np.concatenate([np.eye(3, 4)[None], want_pose])
    want_glob = compose_global(want_rel)
    for rank, shape, rel_sum, glob_sum, pair_ids, starts in got:
        assert shape == (n_frames, 3, 4) and np.array_equal(pair_ids, np.arange(n_frames - 1))
        assert rel_sum == float(np.abs(want_rel).sum()) and glob_sum == float(np.abs(want_glob).sum())
        assert starts.tolist() == [432 * r for r in range(8)]          # each pair came from the rank that owns it


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


# ---- loop-closure candidate validation: the flat (candidate, current) pair list split over the ranks ----
def _loop_case(n_pairs, n_frames=40):
    rng = np.random.default_rng(9)
    f1 = np.sort(rng.integers(1, n_frames, n_pairs)).astype(np.int32)                  # current frames, frame by frame
    f0 = np.array([rng.integers(0, f) for f in f1], np.int32)                          # an earlier candidate each
    seeds = rng.integers(0, 2 ** 32, n_pairs, dtype=np.uint64)
    return f0, f1, seeds


def _fake_block(frames):
    """Stand-in for the device pass: a pair's results are a function of its two MOVIE frames and its seed only, so a
    block that mis-translates frame positions or seeds shows up in the gathered arrays."""
    def run(used, l0, l1, seeds, stereo_key=None):
        g0, g1 = frames[used][l0], frames[used][l1]
        p = len(l0)
        pose = np.array([_fake_pose(int(a) * 1000 + int(b)) for a, b in zip(g0, g1)]).reshape(p, 3, 4)
        stats = np.zeros((p, 8), np.int32)
        stats[:, 0], stats[:, 1], stats[:, 2], stats[:, 3] = g0, g1, (seeds & 0x7FFFFFFF).astype(np.int32), len(used)
        if stereo_key is not None:                                                     # two phases: the used frames' key rows
            sk = stereo_key.cpu().numpy()
            assert sk.shape[0] == len(used)
            stats[:, 4], stats[:, 5], stats[:, 6] = sk[l0, 0], sk[l1, 1], sk.shape[1]
            stats[:, 7] = (sk[:, 64:] == -1).all()                                     # narrower blocks padded with KEY_NONE
        cnt = ((g0 + 2 * g1) % 4).astype(np.int64)                                     # some pairs without a row
        rows = np.array([[a, b, k] + [int(s) & 0xFFFF] * 9 for a, b, s, c in zip(g0, g1, seeds, cnt) for k in range(c)],
                        np.int32).reshape(-1, 12)
        return pose, stats, rows, cnt
    return run


def _fake_stereo(frames, rank):
    """Stand-in for phase 1: a frame's key row is a function of its MOVIE index; every rank returns its own width."""
    import torch

    def match_block(fa, fb):
        w = 64 * (rank + 1)
        return (torch.from_numpy(frames[fa:fb].astype(np.int32))[:, None] * 1000 + torch.arange(w, dtype=torch.int32)[None]).contiguous()
    return match_block


def _loop_worker(rank, world, port, n_pairs, q, plan="one"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f0, f1, seeds = _loop_case(n_pairs)
    frames = np.arange(100, 140)                                                        # movie indices behind the positions
    mine = seeds if rank == 0 else np.zeros_like(seeds)                                 # only rank 0's seeds may matter
    q.put((rank,) + sdist.track_pairs_sharded(f0, f1, mine, _fake_block(frames), 12, n_frames=len(frames),
                                              stereo_block=_fake_stereo(frames, rank), plan=plan))
    dist.barrier()
    dist.destroy_process_group()


def test_pair_list_shards_gather_to_the_single_process_result():
    """SURVEY 8(e) 'Partitioning', second half: loop-closure validation block-partitions the flat pair list; each rank
    sees only the frames its block touches.  Worlds 2 and 3, a ragged last block, and a world with more ranks than
    pairs (a rank with nothing to do)."""
    frames = np.arange(100, 140)
    for world, n_pairs in ((2, 11), (3, 10), (3, 2)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_loop_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        f0, f1, seeds = _loop_case(n_pairs)
        w_pose, w_stats, w_rows, w_cnt = _fake_block(frames)(np.arange(40), f0, f1, seeds)
        for rank, pose, stats, rows, ptr in got:
            assert np.array_equal(pose, w_pose) and np.array_equal(stats[:, :3], w_stats[:, :3])
            assert np.array_equal(rows, w_rows) and np.array_equal(ptr, np.concatenate([[0], np.cumsum(w_cnt)]))
            # each rank handed its block only the frames that block touches
            for r in range(world):
                a, b = sdist.shard_list(n_pairs, r, world)
                if b > a:
                    assert (stats[a:b, 3] == len(np.unique(np.concatenate([f0[a:b], f1[a:b]])))).all()
    used, l0, l1 = sdist.frames_of_block(np.array([5, 9, 5]), np.array([7, 5, 30]), 0, 3)
    assert used.tolist() == [5, 7, 9, 30] and l0.tolist() == [0, 2, 0] and l1.tolist() == [1, 0, 3]


def test_pair_list_in_two_phases_hands_every_block_its_frames_stereo_keys():
    """The two-phase plan of the pair-list split: every frame is stereo-matched on ONE rank (block partition
    of the frame list), the keys are all-gathered -- the path's one exchange step -- and each pair block receives the key
    rows of exactly the frames it touches, in their order.  Ranks return blocks of different widths (padded with
    SLAM_KEY_NONE to the widest); worlds 2 and 3 and a world with a rank without pairs.  Everything else must equal the
    one-phase result."""
    frames = np.arange(100, 140)
    for world, n_pairs in ((2, 11), (3, 10), (3, 2)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_loop_worker, args=(r, world, port, n_pairs, q, "two")) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        f0, f1, seeds = _loop_case(n_pairs)
        w_pose, w_stats, w_rows, w_cnt = _fake_block(frames)(np.arange(40), f0, f1, seeds)
        for rank, pose, stats, rows, ptr in got:
            assert np.array_equal(pose, w_pose) and np.array_equal(stats[:, :3], w_stats[:, :3])
            assert np.array_equal(rows, w_rows) and np.array_equal(ptr, np.concatenate([[0], np.cumsum(w_cnt)]))
            assert np.array_equal(stats[:, 4], frames[f0] * 1000) and np.array_equal(stats[:, 5], frames[f1] * 1000 + 1)
            assert (stats[:, 6] == 64 * world).all()


def test_pair_list_plan_picks_two_phases_only_when_the_blocks_share_frames():
    rng = np.random.default_rng(3)
    F = 400
    cur = np.repeat(np.arange(1, F), 5)
    spread = np.array([rng.integers(0, c) for c in cur])                                # candidates anywhere in the past
    two, one_cost, two_cost = sdist.pair_list_plan(spread, cur, F, 8)
    assert two and two_cost < one_cost
    f0 = np.arange(F - 1)                                                                # consecutive pairs: a halo frame per block
    two, one_cost, two_cost = sdist.pair_list_plan(f0, f0 + 1, F, 8)
    assert not two and one_cost == 50 + 51 and two_cost == 50 + 50
    assert sdist.pair_list_plan(spread, cur, F, 1)[0] is False                          # one rank: nothing to share

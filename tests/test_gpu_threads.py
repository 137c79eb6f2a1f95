"""slam_track_pairs called from several host threads at once (ctypes releases the GIL for the call): every thread has its
own tracker, workspaces and HIP stream; inside the library they share the host worker pool and each has its own
thread-local streams and events (csrc/pipeline.hip: ThreadState).  Results must be what one thread alone gets, and
short-lived threads must be able to come and go (their streams and events are released when they exit)."""
import threading

import numpy as np
import pytest
import torch

from slam_amd import pipeline
from slam_amd.config import HotPathConfig
from slam_amd.synth import SynthConfig, SynthSequence

pytestmark = pytest.mark.gpu


def _shard(seed, n_frames=9, n_features=400):
    seq = SynthSequence(SynthConfig(n_frames=n_frames, n_features=n_features, seed=seed, spawn_per_frame=40, ragged=19))
    desc, xy, count = seq.frames(0, n_frames)
    F, _, n_max, _ = desc.shape
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return F, n_max, (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))


def _same(a, b):
    return (np.array_equal(a.pose, b.pose, equal_nan=True) and np.array_equal(a.stats, b.stats) and np.array_equal(a.mask, b.mask)
            and torch.equal(a.tracked_kp, b.tracked_kp))


def test_trackers_on_concurrent_host_threads_return_the_single_thread_results():
    n_threads, rounds = 4, 12
    jobs = []
    for k in range(n_threads):
        F, n_max, d = _shard(100 + k)
        f0, f1 = pipeline.consecutive_pairs(F)
        trk = pipeline.ShardTracker(F, len(f0), n_max, HotPathConfig(), n_chunks=1 + k % 2)
        seeds = np.arange(len(f0)) + 10 * k
        want = trk.run(*d, f0, f1, seeds)
        want.tracked_kp = want.tracked_kp.clone()
        jobs.append((trk, d, f0, f1, seeds, want))
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(n_threads)

    def worker(k):
        try:
            trk, d, f0, f1, seeds, want = jobs[k]
            stream = torch.cuda.Stream()
            stream.wait_stream(torch.cuda.default_stream())
            start.wait()
            with torch.cuda.stream(stream):
                for it in range(rounds):
                    got = trk.run(*d, f0, f1, seeds)
                    stream.synchronize()
                    if not _same(got, want):
                        errors.append((k, it))
        except Exception as e:                       # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads) and not errors, errors


def test_short_lived_caller_threads_come_and_go():
    F, n_max, d = _shard(7)
    f0, f1 = pipeline.consecutive_pairs(F)
    trk = pipeline.ShardTracker(F, len(f0), n_max, HotPathConfig(), n_chunks=2)
    seeds = np.arange(len(f0))
    want = trk.run(*d, f0, f1, seeds)
    want.tracked_kp = want.tracked_kp.clone()
    torch.cuda.synchronize()
    out = []

    def once():
        got = trk.run(*d, f0, f1, seeds)
        torch.cuda.synchronize()
        out.append(_same(got, want))

    for _ in range(25):                               # each thread creates its streams / events in the library and exits
        t = threading.Thread(target=once)
        t.start()
        t.join(timeout=120)
        assert not t.is_alive()
    assert out == [True] * 25

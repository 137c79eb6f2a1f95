"""CPU-only checks: the C-ABI library loads and exports every symbol include/slam_amd.h declares; the host-side
random stream reproduces numpy's legacy generator; host helpers.  No compute entry point is called."""
import os
import re

import numpy as np
import pytest

from slam_amd import _lib, ops, pipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    return os.path.exists(_lib.LIB_PATH)


pytestmark = pytest.mark.skipif(not _built(), reason="libslam_amd.so not built (run __graft_entry__.build())")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "slam_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.slam_version()


def test_struct_layouts_match_header():
    import ctypes
    assert ctypes.sizeof(_lib.SlamTrackConfig) == (9 + 12 + 12 + 4) * 8 + 8 * 4 + 8      # ... + 8 int32 + ratio
    assert _lib.SlamTrackConfig.ratio.offset % 8 == 0
    assert ctypes.sizeof(_lib.SlamTrackIO) % 8 == 0


@pytest.mark.parametrize("seed,bound", [(0, 2), (1000, 1400), (1234, 1025), (99, 1024), (7, 3), (2 ** 32 - 1, 65537), (5, 1)])
def test_randint_stream_equals_numpy(seed, bound):
    np.random.seed(seed)
    want = np.random.randint(0, bound, 4096)
    got, used = ops.randint_host(seed, bound, 4096)
    assert np.array_equal(got, want)
    # stream position: `used` raw words were consumed, so the next raw word of the stream is the one numpy's generator
    # hands out next (a 32-bit draw over the full range takes exactly one raw word, no rejection)
    words = ops.mt19937_words(seed, used + 1)
    assert int(words[used]) == int(np.random.randint(0, 2 ** 32, dtype=np.uint64))


def test_randint_chunked_calls_equal_one_call():
    # the reference draws 4 at a time (utills.py:502); numpy's bounded generator has no carry-over between calls
    np.random.seed(321)
    want = np.concatenate([np.random.randint(0, 777, 4) for _ in range(300)])
    got, _ = ops.randint_host(321, 777, 1200)
    assert np.array_equal(got, want)


def test_randint_empty_raises_like_numpy():
    with pytest.raises(ValueError):
        ops.randint_host(1, 0, 4)


def test_mt19937_words_equal_numpy():
    rs = np.random.RandomState(4242)
    want = rs.randint(0, 2 ** 32, 2000, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(ops.mt19937_words(4242, 2000), want)


def test_mask_word_roundtrip():
    rng = np.random.default_rng(0)
    m = rng.random((3, 130)) < 0.5
    w = ops.bool_to_mask_words(m, 192)
    assert w.shape == (3, 3)
    assert np.array_equal(ops.mask_words_to_bool(w, 130), m)


def test_compose_global_matches_reference(golden):
    glob = pipeline.compose_global(golden["movie_rel"])
    assert np.array_equal(glob, golden["movie_glob"])
    assert np.array_equal(pipeline.camera_centres(glob), golden["traj_centres"])


def test_ops_refuse_cpu_tensors():
    import torch
    with pytest.raises(ValueError):
        ops.match_keys(torch.zeros((4, 64), dtype=torch.uint8), torch.zeros((4, 64), dtype=torch.uint8))


def test_track_pairs_rejects_bad_arguments_before_touching_the_gpu():
    """Argument validation of slam_track_pairs runs on the host before any HIP call: error code + message, no compute."""
    import ctypes
    L = _lib.lib()
    cfg = _lib.SlamTrackConfig()
    io = _lib.SlamTrackIO()

    def call():
        rc = L.slam_track_pairs(ctypes.byref(cfg), ctypes.byref(io), None)
        return rc, L.slam_last_error_string().decode()

    assert L.slam_track_pairs(None, None, None) == _lib.SLAM_E_INVALID
    io.n_frames, io.n_pairs, io.n_max = 4, 3, 100                     # n_max not a multiple of 64
    rc, msg = call()
    assert rc == _lib.SLAM_E_INVALID and "multiple of 64" in msg
    io.n_max = 128                                                    # sizes fine, pointers still null
    rc, msg = call()
    assert rc == _lib.SLAM_E_INVALID and "null pointer" in msg
    io.n_pairs = 0                                                    # nothing to do is not an error
    assert call()[0] == _lib.SLAM_OK
    # pointers set to host dummies (never dereferenced on the device: validation fails first)
    buf = (ctypes.c_int32 * 70000)()
    addr = ctypes.addressof(buf)
    for name in ("d_desc", "d_xy", "d_count", "h_pair_f0", "h_pair_f1", "h_pair_seed", "d_tracked_kp", "h_pose", "h_stats",
                 "h_mask", "d_workspace", "h_workspace"):
        setattr(io, name, addr)
    io.n_frames, io.n_pairs = 70001, 70000
    rc, msg = call()
    assert rc == _lib.SLAM_E_INVALID and "65535" in msg
    io.n_frames, io.n_pairs = 4, 3
    buf[1] = 9                                                        # pair 1 names frame 9 of 4
    rc, msg = call()
    assert rc == _lib.SLAM_E_INVALID and "pair 1" in msg and "outside" in msg
    with pytest.raises(_lib.SlamError):
        _lib.check(rc)
    # the variant that takes the frames' stereo keys validates the same way, plus the key pointer's alignment
    rc = L.slam_track_pairs_stereo_keys(ctypes.byref(cfg), ctypes.byref(io), addr + 2, None)
    assert rc == _lib.SLAM_E_INVALID and "aligned" in L.slam_last_error_string().decode()
    rc = L.slam_track_pairs_stereo_keys(ctypes.byref(cfg), ctypes.byref(io), addr, None)
    assert rc == _lib.SLAM_E_INVALID and "pair 1" in L.slam_last_error_string().decode()
    assert L.slam_track_pairs_stereo_keys(None, None, None, None) == _lib.SLAM_E_INVALID


def test_the_package_never_touches_the_oracle_or_the_reference_tree():
    """The oracle is the checker (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); nothing under slam_amd/
    may import it, put its directory on sys.path, or open the reference checkout at run time."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slam_amd")
    bad = []
    for dirpath, _, files in os.walk(root):
        for f in files:
            if not f.endswith((".py", ".hip", ".cpp", ".h")):
                continue
            text = open(os.path.join(dirpath, f), encoding="utf-8").read()
            for line_no, line in enumerate(text.splitlines(), 1):
                code = line.split("#", 1)[0]
                if re.search(r"\b(import|from)\s+(slam_oracle|c_oracle)\b|[\"']oracle[\"'/]|open\([^)]*/root/reference", code):
                    bad.append(f"{f}:{line_no}: {line.strip()}")
    assert not bad, bad


def _reference_loop(n, status, cnt, prob=0.9999, max_iterations=1e3):
    """utils/utills.py:488-523 literally, over precomputed (solved?, supporters) per draw: returns
    (iterations, draws consumed, best draw, ran out of draws)."""
    outliers_perc, it, d, inl, outl, best, max_supp = 0.99, 0, 0, 0, 0, -1, -1

    def est(o):
        with np.errstate(divide="ignore"):
            return np.log(1 - prob) / np.log(1 - np.power(1 - o, 4))

    while outliers_perc != 0 and it < est(outliers_perc) and it < max_iterations:
        if d >= len(status):
            return it, d, best, True
        k, d = d, d + 1
        if not status[k]:
            continue                                  # :506-507
        num = int(cnt[k])
        if num > max_supp:
            max_supp, best = num, k
        it += 1
        outl += n - num
        inl += num
        outliers_perc = min(outl / (inl + outl), 0.99)
    return it, d, best, False


def test_ransac_replay_random_sequences_incl_estimates_near_the_iteration_count():
    """The replay skips recomputing the loop's estimate where the cached one settles the test by monotonicity (an
    outlier ratio on the far side of the cached ratio from the answer): 400 random supporter sequences whose estimate
    ends anywhere between a handful and the 1000-iteration cap -- drifting, oscillating and step-changing inlier rates,
    so the ratio crosses the cached one in both directions close to where the loop stops -- against the literal loop."""
    import ctypes
    rng = np.random.default_rng(2024)
    L = _lib.lib()
    n, D = 700, 2600
    stops = []
    for case in range(400):
        rate = rng.uniform(0.3, 0.98)                                    # inlier share of a good hypothesis
        good = rng.random(D) < rng.choice([1.0, 0.95, rng.uniform(0.3, 1.0)])
        kind = case % 4
        t = np.arange(D)
        if kind == 1:
            rate = rate * (1 + 0.4 * np.sin(t / rng.uniform(2, 60)))     # oscillating
        elif kind == 2:
            rate = rate * np.where(t < rng.integers(3, 200), rng.uniform(0.2, 1.0), 1.0)   # a step
        elif kind == 3:
            rate = rate * np.linspace(1.0, rng.uniform(0.3, 1.5), D)     # drifting
        cnt = np.clip(np.where(good, n * rate + rng.normal(0, 8, D), rng.integers(0, 12, D)), 0, n).astype(np.int32)
        status = (rng.random(D) > rng.choice([0.0, 0.1, 0.5])).astype(np.uint8)
        want = _reference_loop(n, status, cnt)
        if want[3]:
            continue
        stops.append(want[0])
        for batch in (37, D):
            st = _lib.SlamRansacState()
            L.slam_ransac_state_init(ctypes.byref(st))
            begin = 0
            while not st.finished:
                end = min(begin + batch, D)
                assert L.slam_ransac_replay(ctypes.byref(st), n, status[begin:end].ctypes.data, cnt[begin:end].ctypes.data,
                                            begin, end, 0.9999, 1e3, 0, 0) == 0
                begin = end
            assert (st.it, st.draws, st.best_draw) == want[:3], (case, batch)
    stops = np.array(stops)
    assert len(stops) > 300 and (stops < 20).sum() > 30 and ((stops > 50) & (stops < 1000)).sum() > 30 and (stops == 1000).sum() > 10


@pytest.mark.parametrize("case", ["clean", "half_outliers", "all_fail_then_good", "perfect", "hopeless"])
def test_ransac_replay_follows_the_reference_loop(case):
    """slam_ransac_replay (host only): the loop's bookkeeping over batches of precomputed hypotheses equals the
    reference's loop draw for draw, whatever the batch boundaries; also what slam_track_pairs replays internally."""
    import ctypes
    rng = np.random.default_rng({"clean": 1, "half_outliers": 2, "all_fail_then_good": 3, "perfect": 4, "hopeless": 5}[case])
    n, D = 900, 3000
    status = (rng.random(D) > 0.1).astype(np.uint8)
    if case == "clean":
        cnt = rng.integers(700, 880, D)
    elif case == "half_outliers":
        cnt = np.where(rng.random(D) < 0.3, rng.integers(400, 500, D), rng.integers(0, 30, D))
    elif case == "all_fail_then_good":
        status[:200] = 0
        cnt = rng.integers(600, 900, D)
    elif case == "perfect":
        cnt = np.full(D, n)                            # outliers_perc becomes 0: the loop stops after one iteration
    else:
        cnt = rng.integers(0, 5, D)                    # runs to the 1000-iteration cap
    cnt = np.ascontiguousarray(cnt, dtype=np.int32)
    want = _reference_loop(n, status, cnt)
    assert not want[3]
    L = _lib.lib()
    for batch in (64, 1, 257, D):
        st = _lib.SlamRansacState()
        assert L.slam_ransac_state_init(ctypes.byref(st)) == 0
        begin = 0
        while not st.finished:
            end = min(begin + batch, D)
            rc = L.slam_ransac_replay(ctypes.byref(st), n, status[begin:end].ctypes.data, cnt[begin:end].ctypes.data, begin, end,
                                      0.9999, 1e3, 0, 0)
            assert rc == 0 and st.draws <= end
            begin = end
        assert (st.it, st.draws, st.best_draw, st.hit_cap) == (want[0], want[1], want[2], 0), (case, batch)
    # the bound on draws the reference does not have, and a batch that does not start where the loop stands
    st = _lib.SlamRansacState()
    L.slam_ransac_state_init(ctypes.byref(st))
    zeros = np.zeros(50, np.uint8)
    assert L.slam_ransac_replay(ctypes.byref(st), n, zeros.ctypes.data, cnt.ctypes.data, 0, 50, 0.9999, 1e3, 0, 40) == 0
    assert st.finished and st.hit_cap and st.draws == 40 and st.best_draw == -1
    st = _lib.SlamRansacState()
    L.slam_ransac_state_init(ctypes.byref(st))
    assert L.slam_ransac_replay(ctypes.byref(st), n, zeros.ctypes.data, cnt.ctypes.data, 10, 50, 0.9999, 1e3, 0, 0) == _lib.SLAM_E_INVALID


def test_shutdown_is_idempotent_and_leaves_the_library_usable():
    """slam_shutdown(): joins the host pool (nothing else to release without a GPU); host entry points still work."""
    L = _lib.lib()
    assert L.slam_shutdown() == 0 and L.slam_shutdown() == 0
    out = np.empty(8, np.uint32)
    assert L.slam_mt19937_words(5, out.ctypes.data, 8) == 0
    np.random.seed(5)
    assert np.array_equal(out, np.random.randint(0, 2 ** 32, 8, dtype=np.uint64).astype(np.uint32))

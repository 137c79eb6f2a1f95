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

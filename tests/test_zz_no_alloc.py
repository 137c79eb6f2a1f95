"""slam_match_hamming_noalloc (include/slam_amd.h): the matcher for callers that cannot allocate.  Its partial lists
live in a static device array inside the library; the results must equal slam_match_hamming's bit for bit (the bar for
match lists), on two calls in a row, the second with a smaller input (the array keeps the first call's contents).
Called through the C ABI directly; outputs are preallocated and no torch allocation happens between the two calls."""
import ctypes

import numpy as np
import pytest
import torch

from slam_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _existing(q, t, norm, mode):
    k1, k2 = ops.match_keys(q, t, norm, mode)
    torch.cuda.synchronize()
    return k1.cpu().numpy(), (k2.cpu().numpy() if k2 is not None else None)


def _descs(rng, n, values):
    return torch.from_numpy(ops.pad_descriptors(rng.integers(0, values, (n, 61), dtype=np.uint8))).cuda()


MODES = [_lib.SLAM_MATCH_CROSSCHECK, _lib.SLAM_MATCH_KNN2, _lib.SLAM_MATCH_CROSSCHECK_MUTUAL]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("norm", [_lib.SLAM_NORM_HAMMING2, _lib.SLAM_NORM_HAMMING])
def test_noalloc_equals_existing_entry_point_on_two_calls_second_smaller(norm, mode):
    """All three rules; the mutual one is the only user of the column-minima region of the static array, where tiles
    without rows are never rewritten, so the first call's contents stay behind the second, smaller call."""
    L = _lib.lib()
    rng = np.random.default_rng(77 + norm + 10 * mode)
    # few distinct byte values -> many equal distances -> tie-breaks matter; first call large (many B splits), second small
    shapes = [(3000, 5000), (700, 333)]
    inputs = [(_descs(rng, nq, 4), _descs(rng, nt, 4)) for nq, nt in shapes]
    want = [_existing(q, t, norm, mode) for q, t in inputs]
    outs = [(torch.zeros(nq, dtype=torch.int32, device="cuda"), torch.zeros(nq, dtype=torch.int32, device="cuda"))
            for nq, _ in shapes]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]                       # hipMemGetInfo: the device's free memory
    for (q, t), (k1, k2) in zip(inputs, outs):                       # back to back on one stream, nothing in between
        rc = L.slam_match_hamming_noalloc(_ptr(q), q.shape[0], _ptr(t), t.shape[0], norm, mode, _ptr(k1),
                                          _ptr(k2) if mode == _lib.SLAM_MATCH_KNN2 else None, stream)
        assert rc == _lib.SLAM_OK, L.slam_last_error_string()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before               # the calls took no device memory
    for (k1, k2), (w1, w2) in zip(outs, want):
        assert np.array_equal(k1.cpu().numpy(), w1)
        if mode == _lib.SLAM_MATCH_KNN2:
            assert np.array_equal(k2.cpu().numpy(), w2)
    assert (want[0][0] != -1).any() and (want[1][0] != -1).any()      # the lists are not trivially empty


def test_noalloc_maximum_size_and_rejection_beyond_it():
    L = _lib.lib()
    n = _lib.SLAM_MATCH_NOALLOC_MAX_ROWS
    rng = np.random.default_rng(5)
    q, t = _descs(rng, n, 256), _descs(rng, n, 256)
    t[:500] = q[1000:1500]                                            # real matches
    w1, _ = _existing(q, t, _lib.SLAM_NORM_HAMMING2, _lib.SLAM_MATCH_CROSSCHECK)
    k1 = torch.zeros(n, dtype=torch.int32, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.slam_match_hamming_noalloc(_ptr(q), n, _ptr(t), n, _lib.SLAM_NORM_HAMMING2, _lib.SLAM_MATCH_CROSSCHECK,
                                      _ptr(k1), None, stream)
    assert rc == _lib.SLAM_OK, L.slam_last_error_string()
    torch.cuda.synchronize()
    assert np.array_equal(k1.cpu().numpy(), w1)
    # one row more than the static array covers: refused with a clear message, nothing launched, outputs untouched
    k1.fill_(12345)
    torch.cuda.synchronize()
    for nq, nt in ((n + 1, 10), (10, n + 1)):
        rc = L.slam_match_hamming_noalloc(_ptr(q), nq, _ptr(t), nt, _lib.SLAM_NORM_HAMMING2, _lib.SLAM_MATCH_CROSSCHECK,
                                          _ptr(k1), None, stream)
        assert rc == _lib.SLAM_E_INVALID
        assert str(n + 1).encode() in L.slam_last_error_string() and b"out of range" in L.slam_last_error_string()
    torch.cuda.synchronize()
    assert bool((k1 == 12345).all())


def test_noalloc_empty_inputs():
    L = _lib.lib()
    t = _descs(np.random.default_rng(0), 10, 256)
    k1 = torch.full((10,), 7, dtype=torch.int32, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.slam_match_hamming_noalloc(None, 0, _ptr(t), 10, 7, 0, _ptr(k1), None, stream) == _lib.SLAM_OK
    assert L.slam_match_hamming_noalloc(_ptr(t), 10, None, 0, 7, 0, _ptr(k1), None, stream) == _lib.SLAM_OK
    torch.cuda.synchronize()
    assert bool((k1 == -1).all())                                     # no train rows: every query unmatched


@pytest.mark.parametrize("mode", MODES)
def test_noalloc_calls_can_be_captured_in_a_hip_graph_and_replayed(mode):
    """What the entry point exists for: a caller under HIP graph capture, where hipMalloc / hipFree (and anything else
    that synchronises the device) is an error.  After one warm-up call (which looks the static array's address up), two
    calls are captured into a graph -- capture fails if either allocates -- the outputs are cleared, the graph is
    replayed, and the replayed results must equal slam_match_hamming's."""
    L = _lib.lib()
    norm = _lib.SLAM_NORM_HAMMING2
    rng = np.random.default_rng(300 + mode)
    shapes = [(2500, 4100), (600, 290)]
    inputs = [(_descs(rng, nq, 4), _descs(rng, nt, 4)) for nq, nt in shapes]
    want = [_existing(q, t, norm, mode) for q, t in inputs]
    outs = [(torch.zeros(nq, dtype=torch.int32, device="cuda"), torch.zeros(nq, dtype=torch.int32, device="cuda"))
            for nq, _ in shapes]

    def calls():
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for (q, t), (k1, k2) in zip(inputs, outs):
            rc = L.slam_match_hamming_noalloc(_ptr(q), q.shape[0], _ptr(t), t.shape[0], norm, mode, _ptr(k1),
                                              _ptr(k2) if mode == _lib.SLAM_MATCH_KNN2 else None, stream)
            assert rc == _lib.SLAM_OK, L.slam_last_error_string()

    calls()                                                           # warm-up: hipGetSymbolAddress happens here
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                                     # capture_error_mode "global": an allocation anywhere fails it
        calls()
    for k1, k2 in outs:
        k1.fill_(-7)
        k2.fill_(-7)
    graph.replay()
    torch.cuda.synchronize()
    for (k1, k2), (w1, w2) in zip(outs, want):
        assert np.array_equal(k1.cpu().numpy(), w1)
        if mode == _lib.SLAM_MATCH_KNN2:
            assert np.array_equal(k2.cpu().numpy(), w2)

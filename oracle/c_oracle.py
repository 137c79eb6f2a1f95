"""ctypes front door of the C oracle (oracle/hamming_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_match_crosscheck.restype = ctypes.c_int
        _lib.oracle_match_crosscheck_w64.restype = ctypes.c_int
        _lib.oracle_match_crosscheck_mutual.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def match_crosscheck(q, t, norm=7, mutual=False):
    """(queryIdx, trainIdx, dist) ascending in queryIdx; byte-table code path.  mutual: the newer-OpenCV rule."""
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    nq, nt = len(q), len(t)
    out_t = np.empty(nq, np.int32)
    out_d = np.empty(nq, np.int32)
    fn = lib().oracle_match_crosscheck_mutual if mutual else lib().oracle_match_crosscheck
    fn(_p(q), nq, q.strides[0] if nq else 0, _p(t), nt, t.strides[0] if nt else 0,
       q.shape[1] if nq else 0, norm, _p(out_t), _p(out_d))
    qi = np.nonzero(out_t >= 0)[0].astype(np.int32)
    return qi, out_t[qi], out_d[qi]


def pad64(d):
    d = np.ascontiguousarray(d, dtype=np.uint8)
    out = np.zeros((len(d), 64), np.uint8)
    out[:, :d.shape[1]] = d
    return out


def match_crosscheck_w64(q64, t64, norm=7):
    """Same result, 64-bit popcount code path; rows already padded to 64 B."""
    nq, nt = len(q64), len(t64)
    out_t = np.empty(nq, np.int32)
    out_d = np.empty(nq, np.int32)
    lib().oracle_match_crosscheck_w64(_p(q64), nq, _p(t64), nt, norm, _p(out_t), _p(out_d))
    qi = np.nonzero(out_t >= 0)[0].astype(np.int32)
    return qi, out_t[qi], out_d[qi]


def knn2(q, t, norm=7):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    nq, nt = len(q), len(t)
    idx = np.empty((nq, 2), np.int32)
    dist = np.empty((nq, 2), np.int32)
    lib().oracle_knn2(_p(q), nq, q.strides[0] if nq else 0, _p(t), nt, t.strides[0] if nt else 0,
                      q.shape[1] if nq else 0, norm, _p(idx), _p(dist))
    return idx, dist

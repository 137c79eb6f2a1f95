"""ctypes loader of libslam_amd.so (the C ABI declared in include/slam_amd.h).

There is no CPU fallback: if the library is missing the import fails loudly, and every
compute entry point needs a HIP device.
"""
from __future__ import annotations

import atexit
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLAM_AMD_LIB") or os.path.join(_HERE, "lib", "libslam_amd.so")   # env: tuning builds only

SLAM_NORM_HAMMING = 6
SLAM_NORM_HAMMING2 = 7
SLAM_MATCH_CROSSCHECK = 0
SLAM_MATCH_KNN2 = 1
SLAM_MATCH_CROSSCHECK_MUTUAL = 2
SLAM_KEY_SHIFT = 20
SLAM_KEY_NONE = 0xFFFFFFFF
SLAM_MATCH_NOALLOC_MAX_ROWS = 8192
SLAM_OK, SLAM_E_INVALID, SLAM_E_WORKSPACE, SLAM_E_HIP, SLAM_E_EMPTY, SLAM_E_DUPLICATE, SLAM_E_HOST = 0, -1, -2, -3, -4, -5, -6

SLAM_PAIR_OK, SLAM_PAIR_EMPTY, SLAM_PAIR_FEW, SLAM_PAIR_DRAW_CAP = 1, 2, 4, 8

c_void_p, c_int, c_size_t, c_double, c_uint32, c_long = (ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t,
                                                          ctypes.c_double, ctypes.c_uint32, ctypes.c_long)


class SlamTrackConfig(ctypes.Structure):
    _fields_ = [("K", c_double * 9), ("M1", c_double * 12), ("M2", c_double * 12), ("rec_err", c_double),
                ("acc", c_double), ("prob", c_double), ("max_iterations", c_double),
                ("fixed_iterations", ctypes.c_int32), ("max_draws", ctypes.c_int32), ("norm", ctypes.c_int32),
                ("first_chunk", ctypes.c_int32), ("pipeline_chunks", ctypes.c_int32), ("pipeline_taper_permille", ctypes.c_int32),
                ("match_mode", ctypes.c_int32), ("pipeline_serial_match", ctypes.c_int32), ("ratio", c_double)]


class SlamTrackIO(ctypes.Structure):
    _fields_ = [("d_desc", c_void_p), ("d_xy", c_void_p), ("d_count", c_void_p),
                ("n_frames", ctypes.c_int32), ("n_max", ctypes.c_int32),
                ("h_pair_f0", c_void_p), ("h_pair_f1", c_void_p), ("h_pair_seed", c_void_p),
                ("n_pairs", ctypes.c_int32),
                ("d_tracked_kp", c_void_p),
                ("h_pose", c_void_p), ("h_stats", c_void_p), ("h_mask", c_void_p),
                ("d_workspace", c_void_p), ("d_workspace_bytes", c_size_t),
                ("h_workspace", c_void_p), ("h_workspace_bytes", c_size_t)]


class SlamRansacState(ctypes.Structure):
    _fields_ = [("it", ctypes.c_int32), ("draws", ctypes.c_int32), ("max_supp", ctypes.c_int32), ("best_draw", ctypes.c_int32),
                ("finished", ctypes.c_int32), ("hit_cap", ctypes.c_int32), ("inliers", ctypes.c_int64),
                ("outliers", ctypes.c_int64), ("outliers_perc", c_double), ("est_for", c_double), ("est", c_double)]


# name -> (restype, argtypes); must list every symbol include/slam_amd.h declares
SIGNATURES = {
    "slam_version": (ctypes.c_char_p, []),
    "slam_last_error_string": (ctypes.c_char_p, []),
    "slam_shutdown": (c_int, []),
    "slam_match_workspace_bytes": (c_size_t, [c_int, c_int]),
    "slam_match_hamming": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_size_t, c_void_p]),
    "slam_match_hamming_noalloc": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "slam_match_batched_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "slam_match_hamming_batched": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_double,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "slam_unpack_keys": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "slam_rect_filter_join": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "slam_triangulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "slam_triangulate_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "slam_mt19937_words": (c_int, [c_uint32, c_void_p, c_int]),
    "slam_randint_host": (c_long, [c_uint32, c_uint32, c_void_p, c_long]),
    "slam_ransac_state_init": (c_int, [c_void_p]),
    "slam_ransac_replay": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_int, c_int]),
    "slam_randint_draws": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "slam_p3p_hypotheses": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p]),
    "slam_score_hypotheses": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                      c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "slam_score_pack_points": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "slam_score_hypotheses_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                             c_int, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p]),
    "slam_consensus_match_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double,
                                         c_void_p, c_void_p]),
    "slam_refit_pnp": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    "slam_compose_global": (c_int, [c_void_p, c_int, c_void_p]),
    "slam_unpack_detections": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.c_int64, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "slam_chain_tracks": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "slam_track_device_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_void_p]),
    "slam_track_host_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_void_p]),
    "slam_track_pairs": (c_int, [c_void_p, c_void_p, c_void_p]),
    "slam_track_pairs_stereo_keys": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "slam_track_last_timing": (c_int, [c_void_p]),
}

_lib = None


class SlamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libslam_amd error {code}: {msg}")
        self.code = code


def lib():
    """The loaded library with typed entry points.  Raises if it has not been built
    (`python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with __graft_entry__.build(); "
                              "slam_amd has no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        missing = [n for n in SIGNATURES if not hasattr(L, n)]
        if missing:
            raise ImportError(f"{LIB_PATH} does not export {missing}: rebuild it with __graft_entry__.build()")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        # Explicit shutdown BEFORE interpreter / library finalisation: joins the host worker pool and releases the main
        # thread's streams and events while the HIP runtime (and a profiler's tool library, if one is attached) is
        # still up, so nothing of this library is left for process-exit handlers to trip over.
        atexit.register(_shutdown)
    return _lib


def _shutdown():
    if _lib is not None:
        _lib.slam_shutdown()


def check(code):
    if code < 0:
        raise SlamError(code, lib().slam_last_error_string().decode())
    return code

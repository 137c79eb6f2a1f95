"""Torch-tensor front door of libslam_amd.so.

PyTorch is plumbing here: it owns device memory and the HIP stream; every computation happens in the
hand-written HIP kernels behind the C ABI (include/slam_amd.h).  There is no CPU path: tensors must live on
a HIP device.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import (SLAM_KEY_NONE, SLAM_KEY_SHIFT, SLAM_MATCH_CROSSCHECK, SLAM_MATCH_CROSSCHECK_MUTUAL, SLAM_MATCH_KNN2,
                   SLAM_NORM_HAMMING, SLAM_NORM_HAMMING2, check)

IDX_MASK = (1 << SLAM_KEY_SHIFT) - 1


def _dev(t: torch.Tensor, dtype=None) -> ctypes.c_void_p:
    if not t.is_cuda:
        raise ValueError("slam_amd ops need tensors on a HIP device (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError("slam_amd ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def _host(a: np.ndarray) -> ctypes.c_void_p:
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f64(a, shape) -> np.ndarray:
    out = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    assert out.shape == shape, (out.shape, shape)
    return out


def _xy_f64(xy1: torch.Tensor) -> int:
    if xy1.dtype == torch.float64:
        return 1
    if xy1.dtype == torch.float32:
        return 0
    raise TypeError("observations must be float32 or float64")


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("slam_amd needs a HIP device (MI355X); none is visible")
    return torch.device("cuda", torch.cuda.current_device())


def pad_descriptors(dsc: np.ndarray) -> np.ndarray:
    """uint8[n, d<=64] -> zero-padded uint8[n, 64] (zero padding is distance-neutral)."""
    dsc = np.asarray(dsc, dtype=np.uint8)
    if dsc.ndim != 2 or dsc.shape[1] > 64:
        raise ValueError("descriptors must be uint8[n, d] with d <= 64")
    if dsc.shape[1] == 64 and dsc.flags["C_CONTIGUOUS"]:
        return dsc
    out = np.zeros((dsc.shape[0], 64), np.uint8)
    out[:, :dsc.shape[1]] = dsc
    return out


# ----------------------------------------------------------------------------------------------- matching
def match_keys(q: torch.Tensor, t: torch.Tensor, norm=SLAM_NORM_HAMMING2, mode=SLAM_MATCH_CROSSCHECK):
    """q, t: uint8[n, 64] on the device.  Returns (key1, key2) int32 tensors holding the packed uint32 keys
    (key2 only for KNN2)."""
    nq, nt = q.shape[0], t.shape[0]
    key1 = torch.empty(nq, dtype=torch.int32, device=q.device)
    key2 = torch.empty(nq, dtype=torch.int32, device=q.device) if mode == SLAM_MATCH_KNN2 else None
    wb = _lib.lib().slam_match_workspace_bytes(nq, nt)
    ws = torch.empty(wb, dtype=torch.uint8, device=q.device)
    check(_lib.lib().slam_match_hamming(_dev(q, torch.uint8), nq, _dev(t, torch.uint8), nt, norm, mode, _dev(key1),
                                        _dev(key2) if key2 is not None else None, _dev(ws), wb, _stream()))
    return key1, key2


def unpack_keys(key: torch.Tensor):
    """packed keys -> (index int32, distance int32); -1 / INT32_MAX where there is no match."""
    flat = key.reshape(-1)
    idx = torch.empty_like(flat)
    dist = torch.empty_like(flat)
    check(_lib.lib().slam_unpack_keys(_dev(flat), flat.numel(), _dev(idx), _dev(dist), _stream()))
    return idx.reshape(key.shape), dist.reshape(key.shape)


def match_keys_batched(desc: torch.Tensor, count: torch.Tensor, prob_q: torch.Tensor, prob_t: torch.Tensor,
                       norm=SLAM_NORM_HAMMING2, mode=SLAM_MATCH_CROSSCHECK, ratio=0.0):
    """desc uint8[n_img, n_max, 64], count int32[n_img], prob_q/prob_t int32[n_prob] -> keys int32[n_prob, n_max].
    ratio > 0 (KNN2 only): key1 keeps only the queries that pass d1 < ratio * d2 (significance_test)."""
    n_img, n_max, _ = desc.shape
    n_prob = prob_q.numel()
    key1 = torch.empty((n_prob, n_max), dtype=torch.int32, device=desc.device)
    key2 = torch.empty((n_prob, n_max), dtype=torch.int32, device=desc.device) if mode == SLAM_MATCH_KNN2 else None
    wb = _lib.lib().slam_match_batched_workspace_bytes(n_prob, n_max, mode)
    ws = torch.empty(wb, dtype=torch.uint8, device=desc.device)
    check(_lib.lib().slam_match_hamming_batched(_dev(desc, torch.uint8), _dev(count, torch.int32), n_max,
                                                _dev(prob_q, torch.int32), _dev(prob_t, torch.int32), n_prob, norm, mode,
                                                float(ratio), _dev(key1), _dev(key2) if key2 is not None else None,
                                                _dev(ws), wb, _stream()))
    return key1, key2


# ----------------------------------------------------------------------------------------- join / triangulate
def rect_filter_join(xy, count, stereo_key, temporal_key, pair_f0, pair_f1, rec_err=2.0):
    """xy float32[n_img, n_max, 2], stereo_key int32[n_frames, n_max], temporal_key int32[n_pairs, n_max]."""
    n_max = xy.shape[1]
    n_pairs = pair_f0.numel()
    dev = xy.device
    tcount = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
    kp = torch.zeros((n_pairs, n_max, 4), dtype=torch.int32, device=dev)
    xy0 = torch.zeros((n_pairs, n_max, 4), dtype=torch.float32, device=dev)
    xy1 = torch.zeros((n_pairs, n_max, 4), dtype=torch.float32, device=dev)
    check(_lib.lib().slam_rect_filter_join(_dev(xy, torch.float32), _dev(count, torch.int32), n_max, _dev(stereo_key),
                                           _dev(temporal_key), _dev(pair_f0, torch.int32), _dev(pair_f1, torch.int32),
                                           n_pairs, float(rec_err), _dev(tcount), _dev(kp), _dev(xy0), _dev(xy1),
                                           _stream()))
    return tcount, kp, xy0, xy1


def triangulate(P, Q, xy0: torch.Tensor, count: torch.Tensor | None):
    """xy0 float32[n_seg, stride, 4] -> X float64[n_seg, stride, 3]."""
    P, Q = _f64(P, (3, 4)), _f64(Q, (3, 4))
    n_seg, stride, _ = xy0.shape
    X = torch.zeros((n_seg, stride, 3), dtype=torch.float64, device=xy0.device)
    check(_lib.lib().slam_triangulate(_host(P), _host(Q), _dev(xy0, torch.float32),
                                      _dev(count, torch.int32) if count is not None else None, n_seg, stride, _dev(X),
                                      _stream()))
    return X


def triangulate_f64(P, Q, xy_l: torch.Tensor, xy_r: torch.Tensor):
    P, Q = _f64(P, (3, 4)), _f64(Q, (3, 4))
    n = xy_l.shape[0]
    X = torch.empty((n, 3), dtype=torch.float64, device=xy_l.device)
    check(_lib.lib().slam_triangulate_f64(_host(P), _host(Q), _dev(xy_l, torch.float64), _dev(xy_r, torch.float64), n,
                                          _dev(X), _stream()))
    return X


# ------------------------------------------------------------------------------------------------- RANSAC
def randint_host(seed: int, bound: int, n: int):
    """np.random.seed(seed); np.random.randint(0, bound, n) -> (values int32[n], raw words consumed)."""
    out = np.empty(n, np.int32)
    used = _lib.lib().slam_randint_host(seed & 0xFFFFFFFF, bound, _host(out), n)
    if used == _lib.SLAM_E_EMPTY:
        raise ValueError("low >= high")
    check(used)
    return out, int(used)


def randint_draws(seed: torch.Tensor, count: torch.Tensor, begin: torch.Tensor, n_draw: int) -> torch.Tensor:
    """Per row r: np.random.seed(seed[r]); draws [begin[r], begin[r] + n_draw) of np.random.randint(0, count[r], 4),
    generated on the device -> int32[n_rows, n_draw, 4]."""
    n_rows = seed.numel()
    out = torch.zeros((n_rows, n_draw, 4), dtype=torch.int32, device=seed.device)
    check(_lib.lib().slam_randint_draws(_dev(seed, torch.int32), _dev(count, torch.int32), _dev(begin, torch.int32), n_rows,
                                        n_draw, _dev(out), _stream()))
    return out


def mt19937_words(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, np.uint32)
    check(_lib.lib().slam_mt19937_words(seed & 0xFFFFFFFF, _host(out), n))
    return out


def p3p_hypotheses(K, X, xy1, count, sample):
    """X float64[n_seg, stride, 3], xy1 float32[n_seg, stride, 4], sample int32[n_seg, n_hyp, 4]
    -> pose float64[n_seg, n_hyp, 3, 4], status uint8[n_seg, n_hyp]."""
    K = _f64(K, (3, 3))
    n_seg, stride, _ = X.shape
    n_hyp = sample.shape[1]
    pose = torch.empty((n_seg, n_hyp, 3, 4), dtype=torch.float64, device=X.device)
    status = torch.empty((n_seg, n_hyp), dtype=torch.uint8, device=X.device)
    check(_lib.lib().slam_p3p_hypotheses(_host(K), _dev(X, torch.float64), _dev(xy1), _xy_f64(xy1),
                                         _dev(count, torch.int32), n_seg, stride, _dev(sample, torch.int32), n_hyp,
                                         _dev(pose), _dev(status), _stream()))
    return pose, status


def score_hypotheses(K, M2, X, xy1, count, pose, status, thr=4.0, want_mask=False):
    K, M2 = _f64(K, (3, 3)), _f64(M2, (3, 4))
    n_seg, stride, _ = X.shape
    n_hyp = pose.shape[1]
    cnt = torch.empty((n_seg, n_hyp), dtype=torch.int32, device=X.device)
    mask = torch.zeros((n_seg, n_hyp, stride // 64), dtype=torch.int64, device=X.device) if want_mask else None
    check(_lib.lib().slam_score_hypotheses(_host(K), _host(M2), _dev(X, torch.float64), _dev(xy1), _xy_f64(xy1),
                                           _dev(count, torch.int32), n_seg, stride, _dev(pose, torch.float64),
                                           _dev(status, torch.uint8), n_hyp, float(thr), _dev(cnt),
                                           _dev(mask) if mask is not None else None, _stream()))
    return cnt, mask


def score_pack_points(X, xy1, count):
    """The points' single-precision records for score_hypotheses_packed: X float64[n_seg, stride, 3], xy1
    float32[n_seg, stride, 4] -> (packed float32[n_seg, stride, 8], scale float32[n_seg])."""
    n_seg, stride, _ = X.shape
    if xy1.dtype != torch.float32:
        raise TypeError("score_pack_points takes float32 observations")
    packed = torch.empty((n_seg, stride, 8), dtype=torch.float32, device=X.device)
    scale = torch.empty(n_seg, dtype=torch.float32, device=X.device)
    check(_lib.lib().slam_score_pack_points(_dev(X, torch.float64), _dev(xy1, torch.float32), _dev(count, torch.int32), n_seg,
                                            stride, _dev(packed), _dev(scale), _stream()))
    return packed, scale


def score_hypotheses_packed(K, M2, X, xy1, packed, scale, count, pose, status, thr=4.0):
    """score_hypotheses' counts through the single-precision pre-classification (include/slam_amd.h)."""
    K, M2 = _f64(K, (3, 3)), _f64(M2, (3, 4))
    n_seg, stride, _ = X.shape
    n_hyp = pose.shape[1]
    cnt = torch.empty((n_seg, n_hyp), dtype=torch.int32, device=X.device)
    check(_lib.lib().slam_score_hypotheses_packed(_host(K), _host(M2), _dev(X, torch.float64), _dev(xy1, torch.float32),
                                                  _dev(packed, torch.float32), _dev(scale, torch.float32),
                                                  _dev(count, torch.int32), n_seg, stride, _dev(pose, torch.float64),
                                                  _dev(status, torch.uint8), n_hyp, float(thr), _dev(cnt), _stream()))
    return cnt


def consensus_match_f64(K, M2, T, X, left1, right1, thr):
    K, M2, T = _f64(K, (3, 3)), _f64(M2, (3, 4)), _f64(T, (3, 4))
    n = X.shape[0]
    mask = torch.empty(n, dtype=torch.uint8, device=X.device)
    check(_lib.lib().slam_consensus_match_f64(_host(K), _host(M2), _host(T), _dev(X, torch.float64),
                                              _dev(left1, torch.float64), _dev(right1, torch.float64), n, float(thr),
                                              _dev(mask), _stream()))
    return mask


def refit_pnp(K, X, xy1, count, mask_words, pose0):
    K = _f64(K, (3, 3))
    n_seg, stride, _ = X.shape
    pose = torch.empty((n_seg, 3, 4), dtype=torch.float64, device=X.device)
    status = torch.empty(n_seg, dtype=torch.uint8, device=X.device)
    check(_lib.lib().slam_refit_pnp(_host(K), _dev(X, torch.float64), _dev(xy1), _xy_f64(xy1), _dev(count, torch.int32),
                                    n_seg, stride, _dev(mask_words, torch.int64), _dev(pose0, torch.float64), _dev(pose),
                                    _dev(status), _stream()))
    return pose, status


def mask_words_to_bool(words: np.ndarray, n: int) -> np.ndarray:
    """uint64/int64 bit words [..., W] -> bool[..., n]."""
    w = np.ascontiguousarray(words).view(np.uint8)
    bits = np.unpackbits(w, axis=-1, bitorder="little")
    return bits[..., :n].astype(bool)


def bool_to_mask_words(mask: np.ndarray, stride: int) -> np.ndarray:
    m = np.zeros(mask.shape[:-1] + (stride,), np.uint8)
    m[..., :mask.shape[-1]] = mask
    return np.packbits(m, axis=-1, bitorder="little").view(np.int64)


def compact_supporters(mask_words: np.ndarray, tracked_count: np.ndarray, tracked_kp: torch.Tensor):
    """The supporters' key-point index rows of a whole shard, compacted ON THE DEVICE and brought back in one pinned
    copy: mask_words int64[P, n_max/64] (host, as slam_track_pairs returns them), tracked_count int[P] (host),
    tracked_kp int32[P, n_max, 4] (device) -> (int32[M, 4] host, int64[P + 1] row pointer).  Same rows, same order as
    slam_amd.tracks.supporters_flat on a host copy; the download is the supporters (16 B each) instead of every
    tracked slot, and goes through page-locked memory (a pageable 113 MB .cpu() ran at 4 GB/s)."""
    dev = tracked_kp.device
    P, n_max = int(tracked_kp.shape[0]), int(tracked_kp.shape[1])
    words = torch.from_numpy(np.ascontiguousarray(mask_words).view(np.int64).reshape(P, n_max // 64)).to(dev)
    T = torch.from_numpy(np.ascontiguousarray(tracked_count, dtype=np.int64)).to(dev)
    bits = ((words.unsqueeze(-1) >> torch.arange(64, device=dev)) & 1).bool().reshape(P, n_max)
    valid = bits & (torch.arange(n_max, device=dev)[None, :] < T[:, None])
    flat_d = tracked_kp[valid]
    if flat_d.shape[0] == 0:
        return np.zeros((0, 4), np.int32), np.zeros(P + 1, np.int64)
    flat_h = torch.empty(flat_d.shape, dtype=torch.int32).pin_memory()
    flat_h.copy_(flat_d, non_blocking=True)
    cnt = valid.sum(1).cpu().numpy()                    # synchronises the stream: the copy above has landed
    ptr = np.zeros(P + 1, np.int64)
    np.cumsum(cnt, out=ptr[1:])
    return flat_h.numpy(), ptr

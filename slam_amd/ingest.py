"""Host -> device ingest of detections: the step in front of the hot path.

The reference detects on the CPU, one frame at a time (utils/utills.py:46-54 `detect_and_compute`, called from :393 and
:839-845), and hands Python lists of cv2.KeyPoint plus an ndarray of descriptors straight to the matcher.  The device
path wants whole shards instead, so the hand-over format is three packed arrays per shard of F frames

    desc  uint8  [2F][n_max][64]   61-byte AKAZE rows (32 for ORB) zero-padded to 64 B; image 2f = left of frame f
    xy    float32[2F][n_max][2]    KeyPoint.pt (cv2 key-points are float32)
    count int32  [2F]              rows in use per image

held in PINNED host memory that the detector workers fill in place (`ShardStager.host_slot`), and moved with one
asynchronous copy per array on a dedicated HIP stream.  Two host slots and two device slots alternate, so shard s+1 is
detected / packed / uploaded while shard s is being tracked; consecutive shards share one halo frame so that every
consecutive frame pair belongs to exactly one shard (same partition as slam_amd/dist.py).

DENSE variant (DenseShardStager): the same shard as descriptor rows at their own width (61 bytes for AKAZE), their
coordinates and a row pointer per image -- 8 % fewer bytes over PCIe, which is what bounds a stream of shards -- that
libslam_amd's slam_unpack_detections expands into the padded device layout (bit-identical to a padded upload).

Nothing here computes on the detections: the arithmetic stays in libslam_amd.  torch is used for pinned memory,
streams and events only.
"""
from __future__ import annotations

import threading
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass

import numpy as np
import torch

from . import ops, pipeline
from .config import HotPathConfig

ROW_BYTES = 64


def round_n_max(n: int) -> int:
    """Per-image slot size: the largest feature count rounded up to the 64 rows the kernels tile by."""
    return max(64, (int(n) + 63) // 64 * 64)


def pack_image(desc_out: np.ndarray, xy_out: np.ndarray, pts, dsc) -> int:
    """Writes one image's detections into its slot (desc_out uint8[n_max,64], xy_out float32[n_max,2]) and returns
    the row count.  pts: sequence of (x, y) or float array [n,2]; dsc: uint8[n, <=64] or None (cv2 returns None for an
    image without key-points, utils/utills.py:52-53).  Rows past the count and bytes past the descriptor width are
    zeroed: the slot is reused and the padding must stay distance-neutral."""
    n = 0 if dsc is None else len(dsc)
    n_max = desc_out.shape[0]
    if n > n_max:
        raise ValueError(f"{n} detections do not fit a slot of {n_max} rows")
    if n:
        d = np.asarray(dsc, dtype=np.uint8)
        if d.ndim != 2 or d.shape[1] > ROW_BYTES:
            raise ValueError("descriptors must be uint8[n, <=64]")
        w = d.shape[1]
        desc_out[:n, :w] = d
        desc_out[:n, w:] = 0
        xy_out[:n] = np.asarray(pts, dtype=np.float32).reshape(n, 2)
    desc_out[n:] = 0
    xy_out[n:] = 0
    return n


@dataclass
class HostSlot:
    desc: np.ndarray     # uint8  [F, 2, n_max, 64]  views of pinned memory
    xy: np.ndarray       # float32[F, 2, n_max, 2]
    count: np.ndarray    # int32  [F, 2]


# A shard goes up in pieces of this many bytes, not as one 0.9 GB copy: the tracker of the shard before it issues small
# host-to-device copies between its kernels (set-up arrays, the winners of a RANSAC round) and waits for them; behind one
# big copy on the same DMA queue each of them would wait for the whole upload.
PIECE_BYTES = int(__import__("os").environ.get("SLAM_INGEST_PIECE_BYTES", 32 << 20))


class ShardStager:
    """Two pinned host slots + two device slots for shards of up to n_frames frames, and the copy stream between them.

        st = ShardStager(F, n_max)
        slot = st.host_slot(k)                 # detector workers write slot.desc / .xy / .count in place
        st.upload(k, n_frames)                 # async H2D on the copy stream (returns at once)
        desc, xy, count = st.device(k)         # the compute stream waits for the copy; tensors are views of slot k
        ... ShardTracker.run(desc, xy, count, ...)
        st.release(k)                          # the next upload into slot k waits for the compute stream

    Device memory: 2 x (2F x n_max x 72 B + 8F) - 2 GB for two 3450-frame shards of 2048-row images."""

    def __init__(self, n_frames: int, n_max: int, n_slots: int = 2, piece_bytes: int = PIECE_BYTES):
        if n_max % 64:
            raise ValueError("n_max must be a multiple of 64")
        self.n_frames, self.n_max, self.n_slots = n_frames, n_max, n_slots
        self.piece_bytes = int(piece_bytes)
        self.dev = ops.device()
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        F = n_frames
        self._h, self._d, self._views = [], [], []
        for _ in range(n_slots):
            h = (torch.zeros((2 * F, n_max, ROW_BYTES), dtype=torch.uint8).pin_memory(),
                 torch.zeros((2 * F, n_max, 2), dtype=torch.float32).pin_memory(),
                 torch.zeros((2 * F,), dtype=torch.int32).pin_memory())
            d = tuple(torch.empty(t.shape, dtype=t.dtype, device=self.dev) for t in h)
            self._h.append(h)
            self._d.append(d)
            self._views.append(HostSlot(h[0].numpy().reshape(F, 2, n_max, ROW_BYTES), h[1].numpy().reshape(F, 2, n_max, 2),
                                        h[2].numpy().reshape(F, 2)))
        self._uploaded = [torch.cuda.Event() for _ in range(n_slots)]
        self._released = [None] * n_slots
        self._frames = [0] * n_slots

    def host_slot(self, k: int) -> HostSlot:
        return self._views[k % self.n_slots]

    def bytes_per_shard(self, n_frames: int | None = None) -> int:
        F = self.n_frames if n_frames is None else n_frames
        return 2 * F * self.n_max * (ROW_BYTES + 8) + 2 * F * 4

    def upload(self, k: int, n_frames: int | None = None) -> None:
        k %= self.n_slots
        F = self.n_frames if n_frames is None else int(n_frames)
        if not 0 < F <= self.n_frames:
            raise ValueError("shard larger than the stager's slots")
        with torch.cuda.stream(self.copy_stream):
            if self._released[k] is not None:
                self.copy_stream.wait_event(self._released[k])     # compute still reading the device slot
            for h, d in zip(self._h[k], self._d[k]):
                rows = max(1, self.piece_bytes // max(1, h[0].numel() * h.element_size()))
                for a in range(0, 2 * F, rows):                    # in pieces: see piece_bytes
                    b = min(2 * F, a + rows)
                    d[a:b].copy_(h[a:b], non_blocking=True)
            self._uploaded[k].record(self.copy_stream)
        self._frames[k] = F

    def device(self, k: int):
        k %= self.n_slots
        F = self._frames[k]
        if F == 0:
            raise RuntimeError("slot was never uploaded")
        torch.cuda.current_stream(self.dev).wait_event(self._uploaded[k])
        d = self._d[k]
        return d[0][:2 * F], d[1][:2 * F], d[2][:2 * F]

    def release(self, k: int) -> None:
        k %= self.n_slots
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self._released[k] = ev

    def host_ready(self, k: int) -> None:
        """Blocks until the last upload of slot k has left the pinned buffer, i.e. the workers may overwrite it."""
        self._uploaded[k % self.n_slots].synchronize()


class DenseHostSlot:
    """Pinned host arrays of one dense shard: desc uint8[capacity, width], xy float32[capacity, 2], count int32[F, 2],
    row_ptr int64[F, 2].  Detector workers call put(frame, side, pts, dsc) from any thread, in any order: an image gets
    the next free run of rows."""

    def __init__(self, desc, xy, count, row_ptr):
        self.desc, self.xy, self.count, self.row_ptr = desc, xy, count, row_ptr
        self.rows_used = 0
        self._lock = threading.Lock()

    def reset(self) -> None:
        self.rows_used = 0
        self.count[:] = 0
        self.row_ptr[:] = 0

    def put(self, frame: int, side: int, pts, dsc) -> int:
        n = 0 if dsc is None else len(dsc)
        with self._lock:
            a = self.rows_used
            if a + n > len(self.desc):
                raise ValueError("dense slot is full")
            self.rows_used = a + n
        if n:
            d = np.asarray(dsc, dtype=np.uint8)
            w = self.desc.shape[1]
            if d.ndim != 2 or d.shape[1] > w:
                if d.ndim == 2 and not d[:, w:].any():
                    d = d[:, :w]                                  # zero padding beyond the slot's width carries nothing
                else:
                    raise ValueError(f"descriptors must be uint8[n, <={w}]")
            self.desc[a:a + n, :d.shape[1]] = d
            self.desc[a:a + n, d.shape[1]:] = 0
            self.xy[a:a + n] = np.asarray(pts, dtype=np.float32).reshape(n, 2)
        self.count[frame, side], self.row_ptr[frame, side] = n, a
        return n


class DenseShardStager(ShardStager):
    """ShardStager whose host slots and uploads are dense (see the module docstring); device(k) unpacks into the same
    padded device tensors ShardStager hands out.

        st = DenseShardStager(F, n_max, width=61)
        slot = st.host_slot(k); slot.reset(); slot.put(frame, side, pts, dsc) ...
        st.upload(k, n_frames); desc, xy, count = st.device(k); ...; st.release(k)"""

    def __init__(self, n_frames: int, n_max: int, width: int = 61, n_slots: int = 2, piece_bytes: int = PIECE_BYTES):
        if n_max % 64:
            raise ValueError("n_max must be a multiple of 64")
        if not 1 <= width <= ROW_BYTES:
            raise ValueError("descriptor width must be 1 .. 64 bytes")
        self.n_frames, self.n_max, self.n_slots, self.width = n_frames, n_max, n_slots, width
        self.piece_bytes = int(piece_bytes)
        self.dev = ops.device()
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        F, cap = n_frames, 2 * n_frames * n_max
        self._h, self._dd, self._d, self._views = [], [], [], []
        for _ in range(n_slots):
            h = (torch.zeros((cap, width), dtype=torch.uint8).pin_memory(), torch.zeros((cap, 2), dtype=torch.float32).pin_memory(),
                 torch.zeros((2 * F,), dtype=torch.int32).pin_memory(), torch.zeros((2 * F,), dtype=torch.int64).pin_memory())
            self._h.append(h)
            self._dd.append(tuple(torch.empty(t.shape, dtype=t.dtype, device=self.dev) for t in h))      # dense, on the device
            self._d.append((torch.empty((2 * F, n_max, ROW_BYTES), dtype=torch.uint8, device=self.dev),
                            torch.empty((2 * F, n_max, 2), dtype=torch.float32, device=self.dev),
                            torch.empty((2 * F,), dtype=torch.int32, device=self.dev)))
            self._views.append(DenseHostSlot(h[0].numpy(), h[1].numpy(), h[2].numpy().reshape(F, 2), h[3].numpy().reshape(F, 2)))
        self._uploaded = [torch.cuda.Event() for _ in range(n_slots)]
        self._released = [None] * n_slots
        self._frames = [0] * n_slots
        self._rows = [0] * n_slots

    def bytes_per_shard(self, n_frames: int | None = None, k: int = 0) -> int:
        F = self.n_frames if n_frames is None else n_frames
        return self._views[k % self.n_slots].rows_used * (self.width + 8) + 2 * F * 12

    def upload(self, k: int, n_frames: int | None = None) -> None:
        k %= self.n_slots
        F = self.n_frames if n_frames is None else int(n_frames)
        if not 0 < F <= self.n_frames:
            raise ValueError("shard larger than the stager's slots")
        rows = self._views[k].rows_used
        with torch.cuda.stream(self.copy_stream):
            if self._released[k] is not None:
                self.copy_stream.wait_event(self._released[k])     # compute still reading the device slot
            for h, d, n in zip(self._h[k], self._dd[k], (rows, rows, 2 * F, 2 * F)):
                step = max(1, self.piece_bytes // max(1, h[0].numel() * h.element_size()))
                for a in range(0, n, step):
                    b = min(n, a + step)
                    d[a:b].copy_(h[a:b], non_blocking=True)
            self._uploaded[k].record(self.copy_stream)
        self._frames[k], self._rows[k] = F, rows

    def device(self, k: int):
        k %= self.n_slots
        F = self._frames[k]
        if F == 0:
            raise RuntimeError("slot was never uploaded")
        torch.cuda.current_stream(self.dev).wait_event(self._uploaded[k])
        dd, d = self._dd[k], self._d[k]
        from . import _lib
        _lib.check(_lib.lib().slam_unpack_detections(dd[0].data_ptr(), dd[1].data_ptr(), dd[3].data_ptr(), dd[2].data_ptr(), 2 * F,
                                                     self.n_max, self.width, self._rows[k], d[0].data_ptr(), d[1].data_ptr(),
                                                     d[2].data_ptr(), ops._stream()))
        return d[0][:2 * F], d[1][:2 * F], d[2][:2 * F]


def fill_dense_slot(slot: DenseHostSlot, source, first: int, last: int, workers: int = 8) -> None:
    """fill_slot for a dense slot: frames [first, last) of a detection source, one frame per task, rows allocated as the
    images arrive."""
    slot.reset()

    def one(j):
        pl, dl, pr, dr = source.get(first + j)
        slot.put(j, 0, pl, dl)
        slot.put(j, 1, pr, dr)
    n = last - first
    if workers <= 1 or n < 4:
        for j in range(n):
            one(j)
    else:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(one, range(n)))


class CachedDetections:
    """Detection source that remembers the last `capacity` frames.

    The reference re-reads both images and re-runs AKAZE for the current frame and for every loop-closure candidate
    on every call (utils/utills.py:1126 -> :386-396), although consecutive current frames share most candidates.
    Detection is a deterministic function of the image, so serving a frame from memory changes no result.  Wraps any
    source with `get(i) -> (pts_l, dsc_l, pts_r, dsc_r)`; least recently used frames are dropped first."""

    def __init__(self, source, capacity: int = 256):
        if capacity < 1:
            raise ValueError("capacity must be positive")
        self.source, self.capacity = source, int(capacity)
        self._store = OrderedDict()
        self._lock = threading.Lock()
        self.hits = self.misses = 0

    def __len__(self):
        return len(self.source)

    def get(self, i):
        i = int(i)
        with self._lock:
            got = self._store.get(i)
            if got is not None:
                self._store.move_to_end(i)
                self.hits += 1
                return got
        got = self.source.get(i)                    # detect outside the lock: workers may miss on different frames
        with self._lock:
            self.misses += 1
            self._store[i] = got
            self._store.move_to_end(i)
            while len(self._store) > self.capacity:
                self._store.popitem(last=False)
        return got

    def clear(self):
        with self._lock:
            self._store.clear()


def shard_ranges(n_frames_total: int, frames_per_shard: int):
    """[(first frame, one past last frame)] with a one-frame halo between neighbours: pairs (f, f+1) of shard s are
    first <= f < last-1, so every consecutive pair of the movie falls in exactly one shard."""
    if frames_per_shard < 2:
        raise ValueError("a shard needs at least two frames")
    out, a = [], 0
    while a < n_frames_total - 1:
        b = min(a + frames_per_shard, n_frames_total)
        out.append((a, b))
        a = b - 1
    return out


def fill_slot(slot: HostSlot, source, first: int, last: int, workers: int = 8) -> None:
    """Packs frames [first, last) of a detection source (`get(i) -> (pts_l, dsc_l, pts_r, dsc_r)`, the shape of
    utils/utills.py:46-54 applied to both images) into a host slot, one frame per task."""
    def one(j):
        pl, dl, pr, dr = source.get(first + j)
        slot.count[j, 0] = pack_image(slot.desc[j, 0], slot.xy[j, 0], pl, dl)
        slot.count[j, 1] = pack_image(slot.desc[j, 1], slot.xy[j, 1], pr, dr)
    n = last - first
    if workers <= 1 or n < 4:
        for j in range(n):
            one(j)
    else:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(one, range(n)))


def track_movie_streamed(source, n_frames_total: int, n_max: int, frames_per_shard: int, cfg: HotPathConfig | None = None,
                         seed_of_pair=None, n_chunks: int = 0, workers: int = 8, dense_width: int | None = None):
    """Whole-movie relative poses (find_features_in_consecutive_frames_whole_movie, utils/utills.py:834-873) for a movie
    that is detected shard by shard: while shard s is tracked on the GPU the host packs and uploads shard s+1.
    seed_of_pair(first_pair, n) -> uint32[n] gives np.random.seed values per pair (default 1000 + pair index).
    dense_width: descriptor bytes per row (61 for AKAZE) to send the shards in the dense wire format.
    Returns (pose float64[n-1,3,4], stats int32[n-1,8])."""
    ranges = shard_ranges(n_frames_total, frames_per_shard)
    Fm = max(b - a for a, b in ranges)
    st = DenseShardStager(Fm, n_max, dense_width) if dense_width else ShardStager(Fm, n_max)
    fill = fill_dense_slot if dense_width else fill_slot
    trk = pipeline.ShardTracker(Fm, Fm - 1, n_max, cfg, n_chunks=n_chunks)
    if seed_of_pair is None:
        seed_of_pair = lambda first, n: 1000 + first + np.arange(n)
    pose = np.full((n_frames_total - 1, 3, 4), np.nan)
    stats = np.zeros((n_frames_total - 1, 8), np.int32)
    with ThreadPoolExecutor(max_workers=1) as feeder:
        def stage(s):
            a, b = ranges[s]
            st.host_ready(s)
            fill(st.host_slot(s), source, a, b, workers)
            return s
        pending = feeder.submit(stage, 0)
        for s, (a, b) in enumerate(ranges):
            pending.result()
            st.upload(s, b - a)
            if s + 1 < len(ranges):
                pending = feeder.submit(stage, s + 1)          # packs the other slot while this shard runs
            d_desc, d_xy, d_count = st.device(s)
            f0, f1 = pipeline.consecutive_pairs(b - a)
            res = trk.run(d_desc, d_xy, d_count, f0, f1, seed_of_pair(a, b - a - 1), unpack_mask=False)
            st.release(s)
            pose[a:b - 1], stats[a:b - 1] = res.pose, res.stats
    return pose, stats

"""Out-of-bounds writes: every device buffer the ops wrappers and the shard tracker hand to the library is carved out of
a larger allocation whose margins hold a byte pattern; after each entry point has run on awkward sizes (1 row, sizes
one off the tile and wavefront multiples, empty and full images, counts that do not fill the stride) the margins must
be untouched.  The caching allocator rounds requests up, so a kernel that writes a few elements past its output would
otherwise go unnoticed.  Results themselves are held to the oracle in tests/test_gpu_parity.py; here only a few cheap
invariants are checked beside the margins."""
import numpy as np
import pytest
import torch

from slam_amd import _lib, ops, pipeline
from slam_amd.config import HotPathConfig
from slam_amd.synth import SynthConfig, SynthSequence

pytestmark = pytest.mark.gpu
PAD = 4096          # bytes either side (keeps the 256-byte alignment the allocator gives)
FILL = 0xA5


class GuardedTorch:
    """Stands in for the `torch` module inside slam_amd.ops / slam_amd.pipeline: device allocations get guarded margins,
    everything else is torch's."""

    def __init__(self):
        self.blocks = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def _carve(self, shape, dtype, device, zero):
        shape = (shape,) if isinstance(shape, int) else tuple(int(s) for s in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty(0, dtype=dtype).element_size()
        body = (nbytes + 255) // 256 * 256
        base = torch.full((body + 2 * PAD,), FILL, dtype=torch.uint8, device=device)
        self.blocks.append((base, nbytes))
        inner = base[PAD:PAD + nbytes]
        if zero:
            inner.zero_()
        return inner.view(dtype).reshape(shape)

    def _alloc(self, zero, *size, dtype=None, device=None, **kw):
        shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
        if device is None or torch.device(device).type != "cuda":
            return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device, **kw)
        return self._carve(shape, dtype or torch.float32, device, zero)

    def empty(self, *size, **kw):
        return self._alloc(False, *size, **kw)

    def zeros(self, *size, **kw):
        return self._alloc(True, *size, **kw)

    def empty_like(self, t, **kw):
        return self._alloc(False, tuple(t.shape), dtype=kw.get("dtype", t.dtype), device=kw.get("device", t.device))

    def zeros_like(self, t, **kw):
        return self._alloc(True, tuple(t.shape), dtype=kw.get("dtype", t.dtype), device=kw.get("device", t.device))

    def check(self, what):
        torch.cuda.synchronize()
        assert self.blocks, what
        for base, nbytes in self.blocks:
            head, tail = base[:PAD], base[PAD + nbytes:]
            assert bool((head == FILL).all()) and bool((tail == FILL).all()), f"{what}: write outside a {nbytes}-byte buffer"
        self.blocks.clear()


@pytest.fixture()
def guard(monkeypatch):
    g = GuardedTorch()
    monkeypatch.setattr(ops, "torch", g)
    monkeypatch.setattr(pipeline, "torch", g)
    return g


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


MODES = (_lib.SLAM_MATCH_CROSSCHECK, _lib.SLAM_MATCH_KNN2, _lib.SLAM_MATCH_CROSSCHECK_MUTUAL)


@pytest.mark.parametrize("shape", [(1, 1), (1, 700), (63, 129), (129, 63), (257, 1000), (2000, 1999), (511, 8200), (8200, 511)])
def test_single_matcher_stays_inside_its_buffers(guard, shape):
    rng = np.random.default_rng(3)
    nq, nt = shape
    q, t = rng.integers(0, 256, (nq, 64), dtype=np.uint8), rng.integers(0, 256, (nt, 64), dtype=np.uint8)
    for mode in MODES:
        for norm in (_lib.SLAM_NORM_HAMMING, _lib.SLAM_NORM_HAMMING2):
            k1, k2 = ops.match_keys(dev(q), dev(t), norm, mode)
            guard.check(f"slam_match_hamming {shape} mode {mode}")
            assert k1.shape == (nq,)


@pytest.mark.parametrize("n_max", [64, 192, 2048])
def test_batched_matcher_stays_inside_its_buffers(guard, n_max):
    rng = np.random.default_rng(4)
    counts = np.array([n_max, 0, 1, n_max - 1, n_max // 2 + 1, 65 % (n_max + 1), n_max], np.int32)
    desc = rng.integers(0, 256, (len(counts), n_max, 64), dtype=np.uint8)
    for n_prob in (1, 5, 23):                       # from "split the scanned image over many workgroups" to "no split"
        pq = rng.integers(0, len(counts), n_prob).astype(np.int32)
        pt = rng.integers(0, len(counts), n_prob).astype(np.int32)
        for mode in MODES:
            k1, _ = ops.match_keys_batched(dev(desc), dev(counts), dev(pq), dev(pt), mode=mode, ratio=0.6 if mode == _lib.SLAM_MATCH_KNN2 else 0.0)
            guard.check(f"slam_match_hamming_batched n_max {n_max} problems {n_prob} mode {mode}")
            idx, _ = ops.unpack_keys(k1)
            guard.check("slam_unpack_keys")
            idx = idx.cpu().numpy()
            for k in range(n_prob):
                assert (idx[k, counts[pq[k]]:] < 0).all() and (idx[k] < max(1, counts[pt[k]])).all()


def _sequence(n_frames, n_features, ragged):
    seq = SynthSequence(SynthConfig(n_frames=max(n_frames, 8), n_features=n_features, seed=11, spawn_per_frame=30, ragged=ragged))
    return seq.frames(0, n_frames)


def test_stages_stay_inside_their_buffers(guard):
    desc, xy, count = _sequence(4, 180, 17)
    F, _, n_max, _ = desc.shape
    count = count.copy()
    count[2] = (0, count[2, 1])                     # a frame without left key-points
    d_desc, d_xy, d_count = dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F))
    fr = torch.arange(F, dtype=torch.int32).cuda()
    skey, _ = ops.match_keys_batched(d_desc, d_count, (2 * fr).int(), (2 * fr + 1).int())
    f0 = torch.arange(F - 1, dtype=torch.int32).cuda()
    tkey, _ = ops.match_keys_batched(d_desc, d_count, (2 * f0).int(), (2 * f0 + 2).int())
    guard.check("matching")
    tcount, kp, xy0, xy1 = ops.rect_filter_join(d_xy, d_count, skey, tkey, f0, f0 + 1, 2.0)
    guard.check("slam_rect_filter_join")
    T = tcount.cpu().numpy()
    assert T[0] > 20 and T[1] == 0 and T[2] == 0
    from slam_amd.synth import K_KITTI, M1_KITTI, M2_KITTI
    X = ops.triangulate(K_KITTI @ M1_KITTI, K_KITTI @ M2_KITTI, xy0, tcount)
    guard.check("slam_triangulate")
    X1 = ops.triangulate_f64(K_KITTI @ M1_KITTI, K_KITTI @ M2_KITTI, xy0[0, :7, :2].double().contiguous(), xy0[0, :7, 2:].double().contiguous())
    guard.check("slam_triangulate_f64")
    assert torch.allclose(X1, X[0, :7], rtol=1e-9, atol=1e-9)
    P = F - 1
    for n_hyp in (1, 64, 300):
        seed = dev(np.arange(P, dtype=np.int32) + 9)
        begin = dev(np.zeros(P, np.int32))
        sample = ops.randint_draws(seed, torch.clamp(tcount, min=1), begin, n_hyp)
        guard.check(f"slam_randint_draws {n_hyp}")
        pose, status = ops.p3p_hypotheses(K_KITTI, X, xy1, tcount, sample)
        guard.check(f"slam_p3p_hypotheses {n_hyp}")
        cnt, _ = ops.score_hypotheses(K_KITTI, M2_KITTI, X, xy1, tcount, pose, status, 4.0)
        guard.check(f"slam_score_hypotheses counts {n_hyp}")
        cnt_m, mask = ops.score_hypotheses(K_KITTI, M2_KITTI, X, xy1, tcount, pose, status, 4.0, want_mask=True)
        guard.check(f"slam_score_hypotheses masks {n_hyp}")
        assert torch.equal(cnt, cnt_m) and int(cnt[1:].abs().sum()) == 0
        packed, scale = ops.score_pack_points(X, xy1, tcount)
        guard.check(f"slam_score_pack_points {n_hyp}")
        cnt_p = ops.score_hypotheses_packed(K_KITTI, M2_KITTI, X, xy1, packed, scale, tcount, pose, status, 4.0)
        guard.check(f"slam_score_hypotheses_packed {n_hyp}")
        assert torch.equal(cnt, cnt_p)
    best = int(cnt[0].argmax())
    words = mask[:, best].contiguous()
    pose1, st = ops.refit_pnp(K_KITTI, X, xy1, tcount, words, pose[:, best].contiguous())
    guard.check("slam_refit_pnp")
    assert int(st[0]) == 1 and int(st[1]) == 0
    m = ops.consensus_match_f64(K_KITTI, M2_KITTI, pose1[0].cpu().numpy(), X[0, :T[0]].contiguous(),
                                xy1[0, :T[0], :2].double().contiguous(), xy1[0, :T[0], 2:].double().contiguous(), 4.0)
    guard.check("slam_consensus_match_f64")
    assert int(m.sum()) >= int(cnt[0, best]) - 5


@pytest.mark.parametrize("chunks", [1, 2, 3])
@pytest.mark.parametrize("method", ["crosscheck", "mutual", "knn"])
def test_shard_tracker_stays_inside_its_workspace(guard, chunks, method):
    """slam_track_pairs carves everything out of ONE device workspace whose size slam_track_device_workspace_bytes
    reports: a carve-up that outgrows the report lands in the margin."""
    from slam_amd.config import REGULAR_KNN
    desc, xy, count = _sequence(7, 300, 23)
    F, _, n_max, _ = desc.shape
    kw = {"crosscheck": {}, "mutual": {"crosscheck_rule": "mutual"}, "knn": {"match_method": REGULAR_KNN}}[method]
    f0, f1 = pipeline.consecutive_pairs(F)
    f0, f1 = np.concatenate([f0, [0, 5]]).astype(np.int32), np.concatenate([f1, [3, 1]]).astype(np.int32)   # + two non-consecutive pairs
    trk = pipeline.ShardTracker(F, len(f0), n_max, HotPathConfig(**kw), n_chunks=chunks)
    d = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
    res = trk.run(*d, f0, f1, np.arange(len(f0)) + 100)
    res2 = trk.run(*d, f0[:3], f1[:3], np.arange(3) + 100)            # a smaller shard in the same workspace
    guard.check(f"slam_track_pairs chunks {chunks} {method}")
    assert res.ok[:F - 1].all() and np.array_equal(res2.pose, res.pose[:3]) and np.array_equal(res2.mask, res.mask[:3])


def test_the_guard_itself_notices_a_stray_write(guard):
    t = guard.empty((10, 3), dtype=torch.int32, device="cuda")
    base, nbytes = guard.blocks[-1]
    assert t.data_ptr() == base.data_ptr() + PAD and nbytes == 120 and t.data_ptr() % 256 == 0
    base[PAD + nbytes] = 0                                            # one byte past the end
    with pytest.raises(AssertionError, match="write outside"):
        guard.check("probe")
    guard.empty(5, dtype=torch.float64, device="cuda")
    guard.blocks[-1][0][PAD - 1] = 0                                  # one byte before the start
    with pytest.raises(AssertionError, match="write outside"):
        guard.check("probe")

"""Host logic of the detection ingest (slam_amd/ingest.py): packing and shard partition.  No GPU."""
import numpy as np
import pytest

from slam_amd import ingest


def _slot(F, n_max):
    return ingest.HostSlot(np.full((F, 2, n_max, 64), 0xAB, np.uint8), np.full((F, 2, n_max, 2), 7.0, np.float32),
                           np.full((F, 2), -1, np.int32))


def test_round_n_max():
    assert [ingest.round_n_max(n) for n in (0, 1, 64, 65, 2000, 2048)] == [64, 64, 64, 128, 2048, 2048]


def test_pack_image_pads_and_clears_a_reused_slot():
    rng = np.random.default_rng(0)
    s = _slot(1, 128)
    d = rng.integers(0, 256, (100, 61), dtype=np.uint8)
    p = rng.random((100, 2)).astype(np.float32) * 1000
    assert ingest.pack_image(s.desc[0, 0], s.xy[0, 0], [tuple(x) for x in p], d) == 100
    assert np.array_equal(s.desc[0, 0, :100, :61], d) and not s.desc[0, 0, :100, 61:].any() and not s.desc[0, 0, 100:].any()
    assert np.array_equal(s.xy[0, 0, :100], p) and not s.xy[0, 0, 100:].any()
    # a smaller image into the same slot leaves nothing of the previous one behind; ORB-width rows pad further
    d2 = rng.integers(0, 256, (10, 32), dtype=np.uint8)
    assert ingest.pack_image(s.desc[0, 0], s.xy[0, 0], p[:10], d2) == 10
    assert np.array_equal(s.desc[0, 0, :10, :32], d2) and not s.desc[0, 0, :10, 32:].any() and not s.desc[0, 0, 10:].any()
    # cv2 returns None descriptors for an image without key-points (utils/utills.py:52-53)
    assert ingest.pack_image(s.desc[0, 0], s.xy[0, 0], [], None) == 0 and not s.desc[0, 0].any() and not s.xy[0, 0].any()


def test_pack_image_rejects_what_does_not_fit():
    s = _slot(1, 64)
    with pytest.raises(ValueError):
        ingest.pack_image(s.desc[0, 0], s.xy[0, 0], np.zeros((65, 2)), np.zeros((65, 61), np.uint8))
    with pytest.raises(ValueError):
        ingest.pack_image(s.desc[0, 0], s.xy[0, 0], np.zeros((4, 2)), np.zeros((4, 65), np.uint8))


@pytest.mark.parametrize("total,per", [(2, 2), (10, 4), (10, 10), (11, 4), (3450, 512), (7, 100)])
def test_shard_ranges_cover_every_pair_once(total, per):
    r = ingest.shard_ranges(total, per)
    pairs = [f for a, b in r for f in range(a, b - 1)]
    assert pairs == list(range(total - 1))
    assert all(2 <= b - a <= per for a, b in r)
    assert all(r[i][1] - 1 == r[i + 1][0] for i in range(len(r) - 1))      # one halo frame


def test_shard_ranges_rejects_single_frame_shards():
    with pytest.raises(ValueError):
        ingest.shard_ranges(10, 1)


def test_fill_slot_matches_the_source_serial_and_threaded():
    from slam_amd.synth import SynthConfig, SynthSequence
    from slam_amd.utills import ArrayDetections
    cfg = SynthConfig(n_frames=6, n_features=150, seed=3, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, 6)
    src = ArrayDetections(desc, xy, count)
    for workers in (1, 4):
        s = _slot(5, cfg.n_max)
        ingest.fill_slot(s, src, 1, 6, workers)
        assert np.array_equal(s.count, count[1:6])
        for j in range(5):
            for side in range(2):
                n = count[1 + j, side]
                assert np.array_equal(s.desc[j, side, :n, :61], desc[1 + j, side, :n, :61])
                assert not s.desc[j, side, :, 61:].any() and not s.desc[j, side, n:].any()
                assert np.array_equal(s.xy[j, side, :n], xy[1 + j, side, :n]) and not s.xy[j, side, n:].any()


def test_cached_detections_serves_repeats_from_memory_and_evicts_lru():
    class Src:
        def __init__(self): self.calls = []
        def __len__(self): return 100
        def get(self, i):
            self.calls.append(i)
            return (np.full((2, 2), i, np.float32), np.full((2, 61), i % 256, np.uint8),) * 2
    src = Src()
    c = ingest.CachedDetections(src, capacity=3)
    assert len(c) == 100
    a = c.get(5); b = c.get(5)
    assert a is b and src.calls == [5] and (c.hits, c.misses) == (1, 1)
    for i in (6, 7):                                   # the loop-closure pattern: same candidates, next current frame
        c.get(i)
    c.get(5)                                           # refreshes 5
    c.get(8)                                           # evicts 6, the least recently used
    assert src.calls == [5, 6, 7, 8]
    c.get(5); c.get(7); c.get(8)
    assert src.calls == [5, 6, 7, 8]
    c.get(6)
    assert src.calls == [5, 6, 7, 8, 6]
    assert np.array_equal(c.get(6)[1], src.get(6)[1])  # what is served is what the source returns
    c.clear(); c.get(8)
    assert src.calls[-2:] == [6, 8]
    with pytest.raises(ValueError):
        ingest.CachedDetections(src, 0)

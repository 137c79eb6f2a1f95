"""Detection ingest (slam_amd/ingest.py) on the GPU: a movie streamed shard by shard through pinned
double buffers gives exactly what one resident shard gives, and one pair across a shard boundary matches the oracle."""
import numpy as np
import pytest
import torch

import c_oracle
import slam_oracle as orc
from slam_amd import ingest
from slam_amd.pipeline import ShardTracker, consecutive_pairs
from slam_amd.synth import SynthConfig, SynthSequence
from slam_amd.utills import ArrayDetections

pytestmark = pytest.mark.gpu


def test_streamed_movie_equals_resident_shard():
    F = 41
    cfg = SynthConfig(n_frames=F, n_features=600, seed=77, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    f0, f1 = consecutive_pairs(F)
    whole = ShardTracker(F, F - 1, n_max).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                              dev(count.reshape(2 * F)), f0, f1, 1000 + np.arange(F - 1), unpack_mask=False)
    src = ArrayDetections(desc, xy, count)
    # 12-frame shards: 4 shards, the last one short, slots reused twice each
    pose, stats = ingest.track_movie_streamed(src, F, n_max, 12, workers=4)
    assert np.array_equal(pose, whole.pose)
    assert np.array_equal(stats[:, :5], whole.stats[:, :5]) and np.array_equal(stats[:, 5], whole.stats[:, 5])
    # the pair that straddles the first shard boundary (frames 11 -> 12 is the first pair of shard 1), against the oracle
    orc.bf_match = lambda a, b, norm=7: c_oracle.match_crosscheck_w64(c_oracle.pad64(a), c_oracle.pad64(b), norm)
    fr = lambda i: dict(dsc_l=desc[i, 0, :count[i, 0], :61], dsc_r=desc[i, 1, :count[i, 1], :61],
                        xy_l=xy[i, 0, :count[i, 0]], xy_r=xy[i, 1, :count[i, 1]])
    for p in (10, 11):
        np.random.seed(1000 + p)
        st = {}
        with np.errstate(all="ignore"):
            want = orc.track_pair(fr(p), fr(p + 1), stats=st)
        assert stats[p, 0] == len(want["kp"]) and stats[p, 1] == int(want["mask"].sum())
        assert stats[p, 2] == st["iterations"]
        assert np.linalg.norm(pose[p] - want["T"]) / np.linalg.norm(want["T"]) <= 1e-5


def test_stager_slots_round_trip_and_reject_oversize():
    st = ingest.ShardStager(4, 128)
    rng = np.random.default_rng(5)
    for k in range(3):                                  # slot 0 is used twice
        s = st.host_slot(k)
        st.host_ready(k)
        s.desc[:] = rng.integers(0, 256, s.desc.shape, dtype=np.uint8)
        s.xy[:] = rng.random(s.xy.shape, dtype=np.float32)
        s.count[:] = rng.integers(0, 129, s.count.shape)
        want = (s.desc.copy(), s.xy.copy(), s.count.copy())
        n = 4 if k != 1 else 3                          # a short shard uploads only its frames
        st.upload(k, n)
        d, x, c = st.device(k)
        assert d.shape[0] == 2 * n
        assert np.array_equal(d.cpu().numpy().reshape(n, 2, 128, 64), want[0][:n])
        assert np.array_equal(x.cpu().numpy().reshape(n, 2, 128, 2), want[1][:n])
        assert np.array_equal(c.cpu().numpy().reshape(n, 2), want[2][:n])
        st.release(k)
    with pytest.raises(ValueError):
        st.upload(0, 5)
    assert st.bytes_per_shard() == 2 * 4 * 128 * 72 + 32


@pytest.mark.parametrize("width", [61, 32, 64])
def test_dense_wire_format_unpacks_to_the_padded_layout(width):
    """DenseShardStager: images put in any order, empty and full ones, several widths; after upload + unpack the
    device tensors equal the padded layout byte for byte (zeros past every count and past the descriptor width), also
    when a slot is reused for a shorter shard that leaves stale rows behind."""
    F, n_max = 5, 192
    st = ingest.DenseShardStager(F, n_max, width=width)
    rng = np.random.default_rng(width)
    for k, frames in enumerate((5, 3, 5)):
        st.host_ready(k)
        slot = st.host_slot(k)
        slot.reset()
        counts = rng.integers(0, n_max + 1, (frames, 2))
        counts[0, 1], counts[frames - 1, 0] = 0, n_max
        want_d = np.zeros((frames, 2, n_max, 64), np.uint8)
        want_x = np.zeros((frames, 2, n_max, 2), np.float32)
        for j, side in rng.permutation([(j, s) for j in range(frames) for s in range(2)]):
            n = int(counts[j, side])
            dsc = rng.integers(0, 256, (n, width), dtype=np.uint8)
            pts = rng.random((n, 2), dtype=np.float32) * 1000
            want_d[j, side, :n, :width], want_x[j, side, :n] = dsc, pts
            assert slot.put(int(j), int(side), pts, dsc if n else None) == n
        st.upload(k, frames)
        d, x, c = st.device(k)
        assert np.array_equal(c.cpu().numpy().reshape(frames, 2), counts)
        assert np.array_equal(d.cpu().numpy().reshape(frames, 2, n_max, 64), want_d)
        assert np.array_equal(x.cpu().numpy().reshape(frames, 2, n_max, 2), want_x)
        assert st.bytes_per_shard(frames, k) == int(counts.sum()) * (width + 8) + 2 * frames * 12
        st.release(k)
    with pytest.raises(ValueError):
        ingest.DenseShardStager(2, 64, width=61).host_slot(0).put(0, 0, np.zeros((300, 2)), np.zeros((300, 61), np.uint8))


def test_unpack_treats_an_image_outside_the_dense_arrays_as_empty():
    from slam_amd import _lib, ops
    n_max, width, rows = 64, 61, 100
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dense = dev(np.full((rows, width), 7, np.uint8))
    xy = dev(np.ones((rows, 2), np.float32))
    big = np.iinfo(np.int64).max
    # 90 + 20 > 100; negative start; 999 clamps to 64, 60 + 64 > 100; a pointer whose `start + rows` wraps around int64
    # (must not pass the range test); the same one short of wrapping; one row past the last admissible start
    row_ptr = dev(np.array([0, 90, -5, 60, big, big - 70, rows - 5 + 1], np.int64))
    count = dev(np.array([10, 20, 3, 999, 64, 64, 5], np.int32))
    n_img = 7
    out_d = torch.full((n_img, n_max, 64), 255, dtype=torch.uint8, device="cuda")
    out_x = torch.full((n_img, n_max, 2), -1.0, dtype=torch.float32, device="cuda")
    out_c = torch.full((n_img,), -1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().slam_unpack_detections(dense.data_ptr(), xy.data_ptr(), row_ptr.data_ptr(), count.data_ptr(), n_img, n_max, width,
                                                 rows, out_d.data_ptr(), out_x.data_ptr(), out_c.data_ptr(), ops._stream()))
    torch.cuda.synchronize()
    assert out_c.cpu().tolist() == [10, 0, 0, 0, 0, 0, 0]
    d = out_d.cpu().numpy()
    assert (d[0, :10, :61] == 7).all() and not d[0, :10, 61:].any() and not d[0, 10:].any() and not d[1:].any()
    assert float(out_x[0, :10].min()) == 1.0 and not out_x[0, 10:].any() and not out_x[1:].any()


def test_streamed_movie_in_the_dense_format_equals_resident_shard():
    F = 29
    cfg = SynthConfig(n_frames=F, n_features=500, seed=78, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    f0, f1 = consecutive_pairs(F)
    whole = ShardTracker(F, F - 1, n_max).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                              dev(count.reshape(2 * F)), f0, f1, 1000 + np.arange(F - 1), unpack_mask=False)
    pose, stats = ingest.track_movie_streamed(ArrayDetections(desc, xy, count), F, n_max, 9, workers=4, dense_width=61)
    assert np.array_equal(pose, whole.pose) and np.array_equal(stats[:, :6], whole.stats[:, :6])

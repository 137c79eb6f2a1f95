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

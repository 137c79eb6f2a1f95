"""BASELINE.json configurations beyond the default: full-size properties (config 2), the stress case (config 4:
8k features, 60 % outliers, fixed hypothesis budget) and batched loop-closure validation (config 5)."""
import numpy as np
import pytest
import torch

import c_oracle
import slam_oracle as orc
from slam_amd import _lib
from slam_amd.config import HotPathConfig
from slam_amd.pipeline import ShardTracker, compose_global, consecutive_pairs
from slam_amd.synth import SynthConfig, SynthSequence

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _frame(desc, xy, count, i):
    n0, n1 = count[i]
    return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])


def _fast_oracle():
    orc.bf_match = lambda a, b, norm=7: c_oracle.match_crosscheck_w64(c_oracle.pad64(a), c_oracle.pad64(b), norm)


def test_full_size_sequence_properties():
    """3450 frames x ~2000 features (BASELINE configs[1]): size-independent properties + oracle spot checks."""
    _fast_oracle()
    F = 3450
    cfg = SynthConfig(n_frames=F, n_features=2000, seed=1234)
    seq = SynthSequence(cfg)
    desc, xy, count = seq.frames(0, F)
    n_max = cfg.n_max
    args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
    f0, f1 = consecutive_pairs(F)
    seeds = 1000 + np.arange(F - 1)
    res = ShardTracker(F, F - 1, n_max, n_chunks=2).run(*args, f0, f1, seeds)
    kp = res.tracked_kp.cpu().numpy()
    T = res.stats[:, 0]
    assert res.ok.all() and (T > 500).all()
    # determinism + independence from the chunking: a second pass, unchunked, is identical bit for bit
    res1 = ShardTracker(F, F - 1, n_max, n_chunks=1).run(*args, f0, f1, seeds)
    assert np.array_equal(res.pose, res1.pose) and np.array_equal(res.mask, res1.mask)
    assert np.array_equal(res.stats[:, :5], res1.stats[:, :5])
    for p in range(0, F - 1, 97):
        k = kp[p, :T[p]]
        assert (np.diff(k[:, 0]) > 0).all()                                  # ascending left0 key-point, no duplicates
        assert len(np.unique(k[:, 2])) == len(k)                              # cross-check: injective into left1
        assert (k[:, 0] < count[p, 0]).all() and (k[:, 1] < count[p, 1]).all()
        assert (k[:, 2] < count[p + 1, 0]).all() and (k[:, 3] < count[p + 1, 1]).all()
        assert (np.abs(xy[p, 0, k[:, 0], 1] - xy[p, 1, k[:, 1], 1]) <= 2).all()   # rectified-stereo test holds
        assert not res.mask[p, T[p]:].any() and res.mask[p, :T[p]].sum() == res.stats[p, 1]
    R = res.pose[:, :, :3]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-9           # proper rotations
    assert (res.stats[:, 2] <= 1000).all() and (res.stats[:, 3] >= res.stats[:, 2]).all()
    assert (res.stats[:, 1] >= 0.8 * T).all()
    # relative poses close to the generator's ground truth; composed trajectory drifts but stays sane
    gt_rel = np.array([seq.gt_relative(i) for i in range(1, F)])
    assert np.abs(res.pose - gt_rel).max() < 0.08
    glob = compose_global(np.concatenate([np.eye(3, 4)[None], res.pose]))
    assert np.isfinite(glob).all()
    # oracle spot checks at full size (bit-exact sets and masks, 1e-5 poses)
    for p in (0, 1723, F - 2):
        np.random.seed(int(seeds[p]))
        st = {}
        with np.errstate(all="ignore"):
            want = orc.track_pair(_frame(desc, xy, count, p), _frame(desc, xy, count, p + 1), stats=st)
        n = len(want["kp"])
        assert T[p] == n and np.array_equal(kp[p, :n], want["kp"])
        assert np.array_equal(res.mask[p, :n], want["mask"])
        assert res.stats[p, 2] == st["iterations"] and res.stats[p, 4] == st["best_draw"]
        assert np.linalg.norm(res.pose[p] - want["T"]) / np.linalg.norm(want["T"]) <= REL_TOL


def test_stress_8k_features_60pct_outliers_10k_hypotheses_full_size():
    """BASELINE configs[3] at full size: 8k features per image, ~60 % outliers among >= 3000 tracked correspondences,
    10 000 counted RANSAC hypotheses per pair (adaptive stop disabled).  Four pairs through the device; properties on
    all of them, the oracle (same 10 000 hypotheses, a few seconds on one core) on the first."""
    import _oracle_pool
    from slam_amd.synth import stress_config
    cfg = stress_config()
    seq = SynthSequence(cfg)
    F, first = 5, 10
    desc, xy, count = seq.frames(first, first + F)
    n_max = cfg.n_max
    hp = HotPathConfig(fixed_iterations=10000)
    f0, f1 = consecutive_pairs(F)
    seeds = 424242 + np.arange(F - 1)
    res = ShardTracker(F, F - 1, n_max, hp).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                                dev(count.reshape(2 * F)), f0, f1, seeds)
    kp = res.tracked_kp.cpu().numpy()
    T, S = res.stats[:, 0], res.stats[:, 1]
    assert count.min() > 7900 and (T >= 3000).all()                            # T ~ 3000-8000
    assert ((1 - S / T) > 0.5).all() and ((1 - S / T) < 0.7).all()             # ~60 % outliers for the winning pose
    assert (res.stats[:, 2] == 10000).all() and (res.stats[:, 3] >= 10000).all() and res.ok.all()
    for p in range(F - 1):
        assert res.mask[p, :T[p]].sum() == S[p] and not res.mask[p, T[p]:].any()
        assert (np.diff(kp[p, :T[p], 0]) > 0).all()
        assert np.abs(res.pose[p] - seq.gt_relative(first + p + 1)).max() < 0.05   # the contaminated set still gives the pose
    import dataclasses
    want = _oracle_pool.run(dataclasses.asdict(cfg), [(0, first, first + 1, seeds[0])], dict(fixed_iterations=10000), workers=1)[0]
    wkp, wmask, wit, wdraws, wbest, wT, _ = want
    n = len(wkp)
    assert T[0] == n and np.array_equal(kp[0, :n], wkp)
    assert (res.stats[0, 2], res.stats[0, 3], res.stats[0, 4]) == (wit, wdraws, wbest)
    assert np.array_equal(np.packbits(res.mask[0, :n]), wmask)
    assert np.linalg.norm(res.pose[0] - wT) / np.linalg.norm(wT) <= REL_TOL


def test_loop_closure_3450_frames_x_5_candidates_full_size():
    """BASELINE configs[4] at full size: every frame of the 3450-frame sequence against 5 candidates (2 revisits + 3
    earlier frames) = 17 220 (candidate, current) pairs in one device pass.  Properties on all pairs, oracle on a sample."""
    import _oracle_pool
    from slam_amd.synth import loop_candidate_pairs
    cfg_kw = dict(n_frames=3450, n_features=2000, seed=1234)
    cfg = SynthConfig(**cfg_kw)
    F = cfg.n_frames
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    pair_f0, pair_f1 = loop_candidate_pairs(F)
    P = len(pair_f0)
    assert P == 5 * (F - 6)
    seeds = 9000 + np.arange(P)
    res = ShardTracker(F, P, n_max).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                        dev(count.reshape(2 * F)), pair_f0, pair_f1, seeds)
    kp = res.tracked_kp.cpu().numpy()
    perc = res.inliers_percentage()
    near = np.arange(P) % 5 < 2
    far = ~near & (np.abs(pair_f1 - pair_f0) > 40)                             # a random candidate may be a revisit too
    assert (perc[near] > 85).mean() > 0.99                                     # revisits pass PoseGraph.py:24's threshold
    assert (perc[far] > 85).sum() == 0 and (perc[far] < 30).mean() > 0.99      # unrelated frames never do
    assert (res.stats[:, 5] & (_lib.SLAM_PAIR_EMPTY | _lib.SLAM_PAIR_FEW | _lib.SLAM_PAIR_OK) != 0).all()
    assert (res.stats[:, 2] <= 1000).all()                                     # the reference's iteration cap (utills.py:491)
    T = res.stats[:, 0]
    assert (res.mask.sum(axis=1) == res.stats[:, 1]).all()
    # revisits (short loops: mostly no rounding-noise draw) and unrelated frames (1000 iterations: nearly always one)
    sample = sorted({*range(0, P, 575), *range(1, P, 1150), *range(2, P, 2300), P - 1})
    want = _oracle_pool.run(cfg_kw, [(p, pair_f0[p], pair_f1[p], seeds[p]) for p in sample])
    clean = 0
    for p in sample:
        wkp, wmask, wit, wdraws, wbest, wT, noisy = want[p]
        n = len(wkp)
        assert T[p] == n and np.array_equal(kp[p, :n], wkp)                    # tracked sets: every sampled pair
        w_perc = 100 * int(np.unpackbits(wmask)[:n].sum()) / n
        assert (perc[p] > 85) == (w_perc > 85)                                 # the acceptance decision (utills.py:1141)
        if noisy:
            continue                                                           # rounding-noise draws
        clean += 1
        assert (res.stats[p, 2], res.stats[p, 4]) == (wit, wbest)
        assert np.array_equal(np.packbits(res.mask[p, :n]), wmask)
        if wT is not None:
            assert np.linalg.norm(res.pose[p] - wT) / np.linalg.norm(wT) <= REL_TOL
        else:
            assert not (res.stats[p, 5] & _lib.SLAM_PAIR_OK)
    assert clean >= 10 and len(sample) >= 40


def test_loop_closure_candidates_batched():
    """BASELINE configs[4]: every frame against 5 candidate frames (2 near revisits, 3 unrelated) in one device pass."""
    _fast_oracle()
    cfg = SynthConfig(n_frames=400, n_features=600, seed=17, spawn_per_frame=60, ragged=23)
    seq = SynthSequence(cfg)
    movie = list(range(0, 30)) + list(range(300, 330))          # two stretches of the drive 200 m apart
    F = len(movie)
    fr = [seq.frame(i) for i in movie]
    desc, xy, count = (np.array([f[k] for f in fr]) for k in ("desc", "xy", "count"))
    n_max = cfg.n_max
    rng = np.random.default_rng(3)
    cur, cand = [], []
    for f in range(36, F):                                       # current frames in the second stretch
        for c in (f - 2, f - 5, *rng.integers(0, 30, 3)):        # 2 revisits + 3 frames of the first stretch (utills.py:1128)
            cand.append(int(c))
            cur.append(f)
    pair_f0, pair_f1 = np.array(cand, np.int32), np.array(cur, np.int32)
    P = len(cur)
    seeds = 9000 + np.arange(P)
    res = ShardTracker(F, P, n_max).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                        dev(count.reshape(2 * F)), pair_f0, pair_f1, seeds)
    kp = res.tracked_kp.cpu().numpy()
    perc = res.inliers_percentage()
    near = np.arange(P) % 5 < 2
    assert (perc[near & (res.stats[:, 0] > 30)] > 85).mean() > 0.9            # revisits pass the reference's threshold (PoseGraph.py:24)
    # unrelated frames still yield ~25 % spurious 4-view tracks (the cross-check rule is not mutual); RANSAC rejects them
    assert (perc[~near] < 30).all()
    assert (res.stats[:, 5] & (_lib.SLAM_PAIR_EMPTY | _lib.SLAM_PAIR_FEW | _lib.SLAM_PAIR_OK) != 0).all()
    for p in (0, 1, 56, P - 4):
        np.random.seed(int(seeds[p]))
        with np.errstate(all="ignore"):
            want = orc.track_pair(_frame(desc, xy, count, pair_f0[p]), _frame(desc, xy, count, pair_f1[p]))
        n = len(want["kp"])
        assert res.stats[p, 0] == n and np.array_equal(kp[p, :n], want["kp"])
        assert np.array_equal(res.mask[p, :n], want["mask"])
        assert np.linalg.norm(res.pose[p] - want["T"]) / np.linalg.norm(want["T"]) <= REL_TOL


def test_degenerate_frames_empty_and_tiny():
    """Edge cases the reference handles by raising / returning None: a frame without features (tracked set empty ->
    ValueError at utills.py:502) and a frame with 3 features (fewer than 6 supporters -> pose None, :558-559)."""
    _fast_oracle()
    cfg = SynthConfig(n_frames=40, n_features=300, seed=2, spawn_per_frame=40, ragged=13)
    seq = SynthSequence(cfg)
    F = 6
    desc, xy, count = seq.frames(0, F)
    count = count.copy()
    count[2] = 0                      # frame 2: nothing detected
    count[4] = 3                      # frame 4: three features per image
    n_max = cfg.n_max
    f0, f1 = consecutive_pairs(F)
    seeds = 50 + np.arange(F - 1)
    res = ShardTracker(F, F - 1, n_max).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                            dev(count.reshape(2 * F)), f0, f1, seeds)
    st = res.stats
    assert st[0, 5] & _lib.SLAM_PAIR_OK
    for p in (1, 2):                                              # pairs touching the empty frame
        assert st[p, 0] == 0 and st[p, 5] & _lib.SLAM_PAIR_EMPTY and np.isnan(res.pose[p]).all() and not res.mask[p].any()
    for p in (3, 4):                                              # pairs touching the 3-feature frame
        assert st[p, 0] <= 3 and np.isnan(res.pose[p]).all()
        assert st[p, 5] & (_lib.SLAM_PAIR_FEW | _lib.SLAM_PAIR_EMPTY)
        if st[p, 0] > 0:
            np.random.seed(int(seeds[p]))
            stt = {}
            with np.errstate(all="ignore"):
                want = orc.track_pair(_frame(desc, xy, count, p), _frame(desc, xy, count, p + 1), stats=stt)
            assert want["T"] is None and st[p, 2] == stt["iterations"] == 1000 and st[p, 3] == stt["draws"]
            assert np.array_equal(res.mask[p, :st[p, 0]], want["mask"])
    # the healthy pair is untouched by its neighbours
    np.random.seed(int(seeds[0]))
    with np.errstate(all="ignore"):
        want = orc.track_pair(_frame(desc, xy, count, 0), _frame(desc, xy, count, 1))
    assert np.array_equal(res.mask[0, :st[0, 0]], want["mask"])
    assert np.linalg.norm(res.pose[0] - want["T"]) / np.linalg.norm(want["T"]) <= REL_TOL


def test_image_counts_outside_the_slot_are_clamped_not_trusted():
    """The per-image row counts are a caller-provided device array the host cannot check without a sync: a count
    above n_max or below zero is clamped to the image slot inside the kernels (same results as the clamped count),
    never an out-of-bounds access."""
    cfg = SynthConfig(n_frames=40, n_features=300, seed=4, spawn_per_frame=40, ragged=13)
    desc, xy, count = SynthSequence(cfg).frames(0, 4)
    F, n_max = 4, cfg.n_max
    desc[:, :, :, :61][np.arange(n_max)[None, None, :] >= count[:, :, None]] = 0           # rows past the count: zeros
    bad = count.copy()
    bad[1, 0] = n_max + 100000                  # beyond the slot -> n_max
    bad[2, 1] = -7                              # negative -> 0
    good = bad.copy()
    good[1, 0], good[2, 1] = n_max, 0
    f0, f1 = consecutive_pairs(F)
    seeds = 77 + np.arange(F - 1)
    run = lambda c: ShardTracker(F, F - 1, n_max).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                                      dev(c.reshape(2 * F)), f0, f1, seeds)
    a, b = run(bad), run(good)
    assert np.array_equal(a.stats, b.stats) and np.array_equal(a.mask, b.mask)
    assert np.array_equal(a.pose, b.pose, equal_nan=True)
    assert np.array_equal(a.tracked_kp.cpu().numpy(), b.tracked_kp.cpu().numpy())
    # frame 2 has no right-image rows: both pairs that touch it have an empty tracked set; pair (0, 1) is unaffected
    assert (b.stats[1, 5] & _lib.SLAM_PAIR_EMPTY) and (b.stats[2, 5] & _lib.SLAM_PAIR_EMPTY)
    assert (b.stats[0, 5] & _lib.SLAM_PAIR_OK) and b.stats[0, 0] > 50


def test_results_do_not_depend_on_chunk_plan():
    """pipeline_chunks / pipeline_taper_permille only schedule the work: every plan gives the single-lane result."""
    F = 60
    cfg = SynthConfig(n_frames=F, n_features=500, seed=21, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
    f0, f1 = consecutive_pairs(F)
    seeds = 77 + np.arange(F - 1)
    base = ShardTracker(F, F - 1, n_max).run(*args, f0, f1, seeds)
    for chunks, taper in ((2, 0.45), (3, 0.5), (4, 0.001), (16, 0.9), (29, 1.0), (30, 0.7)):   # 30 > P/2: falls back to one lane
        res = ShardTracker(F, F - 1, n_max, n_chunks=chunks, taper=taper).run(*args, f0, f1, seeds)
        assert np.array_equal(res.pose, base.pose, equal_nan=True) and np.array_equal(res.mask, base.mask), (chunks, taper)
        assert np.array_equal(res.stats[:, :6], base.stats[:, :6]), (chunks, taper)
        assert np.array_equal(res.tracked_kp.cpu().numpy(), base.tracked_kp.cpu().numpy()), (chunks, taper)


@pytest.mark.parametrize("method,rule", [("bf", "ocv_legacy"), ("bf", "mutual"), ("knn", "ocv_legacy")])
def test_stereo_keys_brought_by_the_caller_give_the_results_of_the_call_that_matches_them_itself(method, rule):
    """slam_track_pairs_stereo_keys (phase 2 of the multi-GPU loop-closure split): with the frames' stereo keys
    computed beforehand (pipeline.stereo_keys = slam_match_hamming_batched on every (left, right) pair) the call launches
    the temporal problems alone and must return what it returns when it matches the stereo pairs itself, bit for bit --
    one lane and chunked, consecutive pairs and a loop-closure-shaped list; the right images' descriptors are then not
    read at all (overwritten with noise here)."""
    from slam_amd import pipeline
    from slam_amd.config import BRUTE_FORCE, MUTUAL, OCV_LEGACY, REGULAR_KNN
    hp = HotPathConfig(match_method=BRUTE_FORCE if method == "bf" else REGULAR_KNN,
                       crosscheck_rule=MUTUAL if rule == "mutual" else OCV_LEGACY)
    F = 40
    cfg = SynthConfig(n_frames=F, n_features=500, seed=33, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    d_desc, d_xy, d_count = dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F))
    keys = pipeline.stereo_keys(d_desc, d_count, hp)
    assert keys.shape == (F, n_max) and keys.dtype == torch.int32
    noisy = d_desc.clone()
    noisy[1::2] = torch.randint(0, 256, noisy[1::2].shape, dtype=torch.uint8, device=noisy.device)
    rng = np.random.default_rng(5)
    lists = [consecutive_pairs(F),
             (np.array([int(rng.integers(0, c)) for c in range(5, F) for _ in range(3)], np.int32), np.repeat(np.arange(5, F), 3).astype(np.int32))]
    for f0, f1 in lists:
        seeds = 400 + np.arange(len(f0))
        for chunks in (1, 3):
            base = ShardTracker(F, len(f0), n_max, hp, n_chunks=chunks).run(d_desc, d_xy, d_count, f0, f1, seeds)
            res = ShardTracker(F, len(f0), n_max, hp, n_chunks=chunks).run(noisy, d_xy, d_count, f0, f1, seeds, stereo_key=keys)
            assert np.array_equal(res.pose, base.pose, equal_nan=True) and np.array_equal(res.mask, base.mask)
            assert np.array_equal(res.stats, base.stats) and (base.stats[:, 5] & _lib.SLAM_PAIR_OK).mean() > 0.3
            assert np.array_equal(res.tracked_kp.cpu().numpy(), base.tracked_kp.cpu().numpy())
    with pytest.raises(ValueError):
        ShardTracker(F, F - 1, n_max, hp).run(d_desc, d_xy, d_count, *consecutive_pairs(F), seeds[:F - 1], stereo_key=keys[:-1])


def test_results_as_views_of_the_output_buffers_equal_the_copies():
    """ShardTracker.run(copy=False) hands out the page-locked buffers the library wrote (what bench.py's step consumes);
    same contents as the default copies, and owned by the tracker: the next run rewrites them."""
    F = 12
    cfg = SynthConfig(n_frames=F, n_features=400, seed=8, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
    f0, f1 = consecutive_pairs(F)
    trk = ShardTracker(F, F - 1, n_max)
    a = trk.run(*args, f0, f1, 5 + np.arange(F - 1), unpack_mask=False)
    v = trk.run(*args, f0, f1, 5 + np.arange(F - 1), unpack_mask=False, copy=False)
    assert np.array_equal(a.pose, v.pose, equal_nan=True) and np.array_equal(a.stats, v.stats) and np.array_equal(a.mask, v.mask)
    assert not v.pose.flags.owndata and a.pose.flags.owndata
    kept = v.pose.copy()
    trk.run(*args, f0, f1, 900 + np.arange(F - 1), unpack_mask=False, copy=False)      # other seeds: other winners
    assert np.array_equal(a.pose, kept, equal_nan=True) and not np.array_equal(v.pose, kept, equal_nan=True)


def test_a_pairs_pose_does_not_depend_on_the_size_of_its_launch():
    """The refit's summation order (wavefronts per pair, LDS or global walk) is a function of the pair's own layout,
    never of how many pairs share the launch: round 2 switched both on the launch size (768 / 256 pairs), so a pair's pose
    bits differed between the one-GPU run and an 8-rank or >= 5-chunk run of the same movie.  1100 pairs with ~250
    supporters each: one call, 2 chunks of 550, 5 chunks of 220, and some pairs in launches of their own."""
    F = 40
    cfg = SynthConfig(n_frames=F, n_features=500, seed=33, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
    P = 1100
    f0 = (np.arange(P) % (F - 1)).astype(np.int32)
    f1 = f0 + 1
    seeds = 4000 + np.arange(P)
    base = ShardTracker(F, P, n_max).run(*args, f0, f1, seeds)
    assert base.ok.all() and np.median(base.supporters) > 64
    for chunks in (2, 5):
        res = ShardTracker(F, P, n_max, n_chunks=chunks).run(*args, f0, f1, seeds)
        assert res.timing_ms[7] == chunks
        assert np.array_equal(res.pose, base.pose) and np.array_equal(res.mask, base.mask), chunks
        assert np.array_equal(res.stats[:, :6], base.stats[:, :6]), chunks
    one = ShardTracker(F, 1, n_max)
    for p in (0, 500, 1099):
        res = one.run(*args, f0[p:p + 1], f1[p:p + 1], seeds[p:p + 1])
        assert np.array_equal(res.pose[0], base.pose[p]) and np.array_equal(res.mask[0], base.mask[p]), p


def test_pair_lists_beyond_one_call_are_run_in_blocks(monkeypatch):
    """slam_track_pairs takes at most 65 535 pairs; ShardTracker.run hands a longer list over block by block, each with
    the frame range its pairs touch.  With the block size turned down to 7 the result equals the one-call result, for
    consecutive pairs and for a loop-closure-like list (far-apart and repeated frames), with and without chunks."""
    F = 24
    cfg = SynthConfig(n_frames=F, n_features=420, seed=5, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
    c0, c1 = consecutive_pairs(F)
    rng = np.random.default_rng(9)
    lists = [(c0, c1), (rng.integers(0, F, 31).astype(np.int32), rng.integers(0, F, 31).astype(np.int32))]
    for f0, f1 in lists:
        seeds = 300 + np.arange(len(f0))
        base = ShardTracker(F, len(f0), n_max).run(*args, f0, f1, seeds)
        for chunks in (1, 2):
            with monkeypatch.context() as m:
                m.setattr(ShardTracker, "MAX_PAIRS_PER_CALL", 7)
                trk = ShardTracker(F, len(f0), n_max, n_chunks=chunks)
                res = trk.run(*args, f0, f1, seeds)
            assert trk.d_ws_bytes < ShardTracker(F, len(f0), n_max, n_chunks=chunks).d_ws_bytes      # sized for one block
            assert res.timing_ms[7] >= -(-len(f0) // 7)                                              # one or more launches per block
            assert np.array_equal(res.pose, base.pose, equal_nan=True) and np.array_equal(res.mask, base.mask)
            assert np.array_equal(res.stats[:, :6], base.stats[:, :6])
            assert np.array_equal(res.tracked_kp.cpu().numpy(), base.tracked_kp.cpu().numpy())


def test_seventy_thousand_pairs_in_one_run():
    """The real limit: 70 000 loop-closure-like pairs over a 40-frame sequence of small images go through in two
    blocks; a sample of them equals the same pairs run on their own."""
    F, P = 40, 70000
    cfg = SynthConfig(n_frames=F, n_features=60, seed=2, spawn_per_frame=10, ragged=5)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)))
    rng = np.random.default_rng(1)
    f0 = rng.integers(0, F, P).astype(np.int32)
    f1 = np.clip(f0 + rng.integers(-2, 3, P), 0, F - 1).astype(np.int32)
    seeds = 5 + np.arange(P)
    res = ShardTracker(F, P, n_max).run(*args, f0, f1, seeds)
    assert res.timing_ms[7] >= 2 and (res.stats[:, 5] != 0).all()                 # every pair has a status: none was skipped
    pick = np.concatenate([np.arange(0, 40), np.arange(65515, 65555), np.arange(P - 40, P)])
    sub = ShardTracker(F, len(pick), n_max).run(*args, f0[pick], f1[pick], seeds[pick])
    assert np.array_equal(res.stats[pick, :6], sub.stats[:, :6]) and np.array_equal(res.mask[pick], sub.mask)
    assert np.array_equal(res.pose[pick], sub.pose, equal_nan=True) and (sub.stats[:, 5] & _lib.SLAM_PAIR_OK).sum() > 60


@pytest.mark.parametrize("world", [2, 3, 8])
def test_frame_range_shards_reproduce_the_whole_movie(world):
    """BASELINE configs[2]: the ranks' blocks (slam_amd.dist.shard_frames: pairs [a, b) + one halo frame), run one
    after the other on this GPU with the global pair index as seed, concatenate to exactly the single-GPU result."""
    from slam_amd import dist as sdist
    F = 45
    cfg = SynthConfig(n_frames=F, n_features=400, seed=8, ragged=True)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    f0, f1 = consecutive_pairs(F)
    whole = ShardTracker(F, F - 1, n_max).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                              dev(count.reshape(2 * F)), f0, f1, 1000 + np.arange(F - 1))
    pose, stats, mask, covered = [], [], [], []
    for rank in range(world):
        a, b, fa, fb = sdist.shard_frames(F, rank, world)
        if b == a:
            continue
        n = fb - fa
        g0, g1 = consecutive_pairs(n)
        r = ShardTracker(n, n - 1, n_max, n_chunks=2).run(dev(desc[fa:fb].reshape(2 * n, n_max, 64)), dev(xy[fa:fb].reshape(2 * n, n_max, 2)),
                                                        dev(count[fa:fb].reshape(2 * n)), g0, g1, 1000 + a + np.arange(b - a))
        pose.append(r.pose); stats.append(r.stats); mask.append(r.mask); covered += list(range(a, b))
    assert covered == list(range(F - 1))
    assert np.array_equal(np.concatenate(pose), whole.pose) and np.array_equal(np.concatenate(mask), whole.mask)
    assert np.array_equal(np.concatenate(stats)[:, :6], whole.stats[:, :6])


def test_draw_cap_bounds_the_retry_loop():
    """The reference's `while` loop has no bound on draws (utills.py:498-523); max_draws is this library's bound.  With a
    budget of 500 iterations and a cap of 100 draws the loop stops at the cap, flags SLAM_PAIR_DRAW_CAP, and returns
    the best of exactly those draws - the oracle with the same cap agrees."""
    _fast_oracle()
    F = 3
    cfg = SynthConfig(n_frames=F, n_features=500, seed=12)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    hp = HotPathConfig(fixed_iterations=500, max_draws=100)
    f0, f1 = consecutive_pairs(F)
    seeds = np.array([31, 32])
    res = ShardTracker(F, F - 1, n_max, hp).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                                dev(count.reshape(2 * F)), f0, f1, seeds)
    for p in range(F - 1):
        np.random.seed(int(seeds[p]))
        st = {}
        with np.errstate(all="ignore"):
            want = orc.track_pair(_frame(desc, xy, count, p), _frame(desc, xy, count, p + 1), stats=st,
                                  fixed_iterations=500, max_draws=100)
        assert res.stats[p, 5] & _lib.SLAM_PAIR_DRAW_CAP and res.stats[p, 5] & _lib.SLAM_PAIR_OK
        assert res.stats[p, 3] == 100 == st["draws"] and res.stats[p, 2] == st["iterations"] <= 100
        assert res.stats[p, 4] == st["best_draw"]
        n = len(want["kp"])
        assert np.array_equal(res.mask[p, :n], want["mask"])
        assert np.linalg.norm(res.pose[p] - want["T"]) / np.linalg.norm(want["T"]) <= REL_TOL
    # without the cap the same pairs are not flagged
    res2 = ShardTracker(F, F - 1, n_max, HotPathConfig(fixed_iterations=500)).run(
        dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)), f0, f1, seeds)
    assert not (res2.stats[:, 5] & _lib.SLAM_PAIR_DRAW_CAP).any() and (res2.stats[:, 2] == 500).all()


def test_workspace_too_small_is_an_error_not_a_fault():
    """slam_track_pairs never allocates: a workspace below slam_track_*_workspace_bytes is SLAM_E_WORKSPACE with the
    needed sizes in the message, and nothing is launched."""
    import ctypes
    F = 4
    cfg = SynthConfig(n_frames=F, n_features=200, seed=2)
    desc, xy, count = SynthSequence(cfg).frames(0, F)
    n_max = cfg.n_max
    trk = ShardTracker(F, F - 1, n_max)
    trk.d_ws_bytes //= 2                                  # lie about the size; the buffer itself stays large enough
    f0, f1 = consecutive_pairs(F)
    with pytest.raises(_lib.SlamError) as e:
        trk.run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)), f0, f1,
                np.arange(F - 1))
    assert "workspace too small" in str(e.value)
    torch.cuda.synchronize()
    good = ShardTracker(F, F - 1, n_max).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                             dev(count.reshape(2 * F)), f0, f1, np.arange(F - 1))
    assert good.ok.all()                                  # the library is still usable after the refusal

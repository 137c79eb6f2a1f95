"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors.

Bars: bit-exact for match lists, joins, sample indices and inlier masks/counts; 1e-5 relative (north_star) for
triangulated points and poses (written next to each assert; most are far tighter in practice)."""
import numpy as np
import pytest
import torch

import c_oracle
import slam_oracle as orc
from slam_amd import _lib, ops
from slam_amd.synth import SynthConfig, SynthSequence

pytestmark = pytest.mark.gpu

K, M1, M2 = orc.K, orc.M1, orc.M2
REL_TOL = 1e-5


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def gpu_crosscheck(q, t, norm, mutual=False):
    mode = _lib.SLAM_MATCH_CROSSCHECK_MUTUAL if mutual else _lib.SLAM_MATCH_CROSSCHECK
    k1, _ = ops.match_keys(dev(ops.pad_descriptors(q)), dev(ops.pad_descriptors(t)), norm, mode)
    idx, dist = ops.unpack_keys(k1)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    qi = np.nonzero(idx >= 0)[0].astype(np.int32)
    return qi, idx[qi], dist[qi]


@pytest.mark.parametrize("mutual", [False, True], ids=["ocv_legacy", "mutual"])
@pytest.mark.parametrize("norm", [orc.NORM_HAMMING2, orc.NORM_HAMMING])
@pytest.mark.parametrize("shape", [(1, 1), (1, 300), (300, 1), (64, 64), (129, 513), (1000, 777), (2000, 1987), (3000, 5000)])
def test_crosscheck_match_bit_exact(norm, shape, mutual):
    """Both cross-check rules cv2.BFMatcher(crossCheck=True) has had (utills.py:57-70; SLAM_MATCH_CROSSCHECK = the
    one-directional legacy rule, SLAM_MATCH_CROSSCHECK_MUTUAL = mutual nearest rows) against the C oracle, on
    tie-heavy rows where first-minimum / lowest-row tie-breaks decide."""
    rng = np.random.default_rng(shape[0] * 7919 + shape[1] + norm)
    nq, nt = shape
    # few distinct byte values -> many equal distances -> tie-breaks matter
    q = rng.integers(0, 4, (nq, 61), dtype=np.uint8)
    t = rng.integers(0, 4, (nt, 61), dtype=np.uint8)
    t[: min(nq, nt) // 2] = q[: min(nq, nt) // 2]          # exact duplicates (distance 0 ties)
    want = c_oracle.match_crosscheck(q, t, norm, mutual=mutual)
    got = gpu_crosscheck(q, t, norm, mutual)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    if mutual and min(shape) >= 64:
        # the mutual rule keeps a subset of the legacy rule's pairs, and strictly fewer on data like this
        legacy = gpu_crosscheck(q, t, norm)
        assert {*zip(got[0].tolist(), got[1].tolist())} < {*zip(legacy[0].tolist(), legacy[1].tolist())}


def test_crosscheck_match_random_full_bytes():
    rng = np.random.default_rng(11)
    q = rng.integers(0, 256, (2000, 61), dtype=np.uint8)
    t = rng.integers(0, 256, (2011, 61), dtype=np.uint8)
    want = c_oracle.match_crosscheck(q, t, orc.NORM_HAMMING2)
    got = gpu_crosscheck(q, t, orc.NORM_HAMMING2)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", [(70000, 6000), (6000, 70000)])
def test_crosscheck_match_row_indices_beyond_16_bits(shape):
    """Row indices above 65535 on either side of the packed keys (distance << 20 | row; SLAM_MAX_ROWS = 2^20): near
    duplicates planted at the far end of the long side must come back with their own indices."""
    rng = np.random.default_rng(shape[0])
    nq, nt = shape
    q = rng.integers(0, 256, (nq, 61), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 61), dtype=np.uint8)
    k = np.arange(200)
    t[nt - 1 - k] = q[nq - 1 - 3 * k]                       # exact copies, far ends of both images
    t[nt - 1 - k, 5] ^= 0x10                                # ... one cell apart
    want = c_oracle.match_crosscheck_w64(c_oracle.pad64(q), c_oracle.pad64(t), orc.NORM_HAMMING2)
    got = gpu_crosscheck(q, t, orc.NORM_HAMMING2)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    hit = {(int(a), int(b)) for a, b, d in zip(*got) if d == 1}
    assert {(nq - 1 - 3 * int(i), nt - 1 - int(i)) for i in k} <= hit
    assert max(max(got[0]), max(got[1])) > 65535


@pytest.mark.parametrize("mutual", [False, True], ids=["ocv_legacy", "mutual"])
def test_match_empty_inputs(mutual):
    q = np.zeros((0, 61), np.uint8)
    t = np.random.default_rng(0).integers(0, 256, (10, 61), dtype=np.uint8)
    assert len(gpu_crosscheck(q, t, orc.NORM_HAMMING2, mutual)[0]) == 0
    assert len(gpu_crosscheck(t, q, orc.NORM_HAMMING2, mutual)[0]) == 0


def test_crosscheck_more_query_rows_than_lds_slots():
    """Above 8192 query rows the legacy rule's per-query slots no longer fit the resolve kernel's LDS array and it
    scatters with global atomics instead: same lists as the oracle on both sides of the limit, both rules."""
    rng = np.random.default_rng(8)
    for nq, nt in ((8192, 3000), (8193, 3000), (9000, 9100)):
        q = rng.integers(0, 256, (nq, 61), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 61), dtype=np.uint8)
        t[:1000] = q[nq - 1000:] ^ (rng.random((1000, 61)) < 0.02).astype(np.uint8)
        for mutual in (False, True):
            want = c_oracle.match_crosscheck_w64(c_oracle.pad64(q), c_oracle.pad64(t), orc.NORM_HAMMING2) if not mutual \
                else c_oracle.match_crosscheck(q, t, orc.NORM_HAMMING2, mutual=True)
            got = gpu_crosscheck(q, t, orc.NORM_HAMMING2, mutual)
            assert len(want[0]) >= 1000
            for a, b in zip(want, got):
                assert np.array_equal(a, b)


@pytest.mark.parametrize("norm", [orc.NORM_HAMMING2, orc.NORM_HAMMING])
@pytest.mark.parametrize("shape", [(5, 2), (200, 333), (1500, 2100)])
def test_knn2_bit_exact(norm, shape):
    rng = np.random.default_rng(shape[0] + 31 * shape[1])
    q = rng.integers(0, 8, (shape[0], 61), dtype=np.uint8)
    t = rng.integers(0, 8, (shape[1], 61), dtype=np.uint8)
    idx_w, dist_w = c_oracle.knn2(q, t, norm)
    k1, k2 = ops.match_keys(dev(ops.pad_descriptors(q)), dev(ops.pad_descriptors(t)), norm, _lib.SLAM_MATCH_KNN2)
    i1, d1 = (x.cpu().numpy() for x in ops.unpack_keys(k1))
    i2, d2 = (x.cpu().numpy() for x in ops.unpack_keys(k2))
    assert np.array_equal(np.stack([i1, i2], 1), idx_w)
    assert np.array_equal(np.stack([d1, d2], 1), dist_w)


@pytest.mark.parametrize("width", [32, 61, 64])
@pytest.mark.parametrize("norm", [orc.NORM_HAMMING2, orc.NORM_HAMMING])
def test_descriptor_widths_orb_akaze_full(norm, width):
    """ORB rows are 32 bytes, AKAZE 61, and a full 64-byte row has no padding at all: same lists as the oracle."""
    rng = np.random.default_rng(width * 10 + norm)
    q = rng.integers(0, 256, (700, width), dtype=np.uint8)
    t = rng.integers(0, 256, (650, width), dtype=np.uint8)
    t[:200] = q[100:300] ^ (rng.random((200, width)) < 0.03).astype(np.uint8)      # near duplicates: real matches
    want = c_oracle.match_crosscheck(q, t, norm)
    got = gpu_crosscheck(q, t, norm)
    assert len(want[0]) > 150
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    idx_w, dist_w = c_oracle.knn2(q, t, norm)
    k1, k2 = ops.match_keys(dev(ops.pad_descriptors(q)), dev(ops.pad_descriptors(t)), norm, _lib.SLAM_MATCH_KNN2)
    i1, d1 = (x.cpu().numpy() for x in ops.unpack_keys(k1))
    i2, d2 = (x.cpu().numpy() for x in ops.unpack_keys(k2))
    assert np.array_equal(np.stack([i1, i2], 1), idx_w) and np.array_equal(np.stack([d1, d2], 1), dist_w)


def _ratio_case():
    """Queries whose two nearest train rows sit at chosen NORM_HAMMING2 distances (d1, d2): the query itself with d1
    and with d2 two-bit cells changed.  (3, 5), (6, 10), (12, 20), (9, 15), (30, 50) are exact ties of
    `d1 < 0.6 * d2`; in float64 every such product is the exact integer for all distances up to 244 (checked by
    enumeration), so ties fail the strict comparison and (2, 5), (29, 50) - one below - pass."""
    rng = np.random.default_rng(606)
    plan = [(3, 5), (6, 10), (3, 5), (12, 20), (2, 5), (4, 5), (0, 7), (9, 15), (30, 50), (29, 50), (5, 5)] * 6
    q = rng.integers(0, 256, (len(plan), 61), dtype=np.uint8)
    t = []
    for row, (d1, d2) in zip(q, plan):
        for d in (d1, d2):
            r = row.copy()
            for cell in rng.choice(244, d, replace=False):          # flip the low bit of d distinct 2-bit cells
                r[cell // 4] ^= 1 << (2 * (cell % 4))
            t.append(r)
    t = np.array(t)
    t = t[rng.permutation(len(t))]
    return q, t, plan


@pytest.mark.parametrize("ratio", [0.6, 0.5, 0.8, 1.0])
def test_knn_ratio_path_matches_oracle_incl_float_ties(ratio):
    """Row 2 end to end through the reference-shaped functions (utills.py:92-102, :131-152): the test
    `d1 < ratio * d2` in float64, strict: exact ties fail, one below the tie passes, as in the reference."""
    from slam_amd import utills as U
    q, t, plan = _ratio_case()
    idx, dist = c_oracle.knn2(q, t, orc.NORM_HAMMING2)
    assert [tuple(d) for d in dist] == [tuple(p) for p in plan]            # the construction delivers the planned distances
    wq, wt, wd = orc.knn_ratio_match(q, t, orc.NORM_HAMMING2, ratio)
    passed, failed = U.significance_test(U.knn_matching(orc.NORM_HAMMING2, q, t), ratio)
    assert len(passed) + len(failed) == len(q) and 0 < len(passed) < len(q)
    assert np.array_equal([m.queryIdx for m in passed], wq) and np.array_equal([m.trainIdx for m in passed], wt)
    assert np.array_equal([m.distance for m in passed], wd)
    if ratio == 0.6:
        got = {m.queryIdx for m in passed}
        assert all((i in got) == (p in ((2, 5), (0, 7), (29, 50))) for i, p in enumerate(plan))


def _small_sequence(n_frames=5, n_features=300, seed=3):
    cfg = SynthConfig(n_frames=40, n_features=n_features, seed=seed, spawn_per_frame=40, ragged=13)
    seq = SynthSequence(cfg)
    desc, xy, count = seq.frames(0, n_frames)
    return seq, desc, xy, count


@pytest.mark.parametrize("mode", ["crosscheck", "knn2", "mutual"])
def test_batched_matcher_rows_per_lane_variants_agree(mode):
    """The batched matcher runs 4 rows per lane when the grid fills the chip, and for a batch too small for that 2 rows
    per lane with the scanned image split over workgroups (a loop-closure call with a handful of candidates,
    utills.py:1118-1146).  The same six problems go through every plan: each alone (most splits), the six together
    (fewer splits), and repeated 70 times in one batch of 420 problems x 2 tiles (4 rows, no split); keys must be
    identical, equal to the single-problem entry point's, and for the cross-check rules equal to the C oracle's."""
    rng = np.random.default_rng(17)
    n_max = 2048
    counts = np.array([1970, 2048, 1, 1025, 700, 0, 1500, 64], np.int32)
    desc = np.zeros((len(counts), n_max, 64), np.uint8)
    for i, n in enumerate(counts):
        desc[i, :n, :61] = rng.integers(0, 6, (n, 61), dtype=np.uint8)      # few byte values: many ties
    desc[1, :900, :61] = desc[0, :900, :61]
    pq = np.array([0, 1, 2, 3, 6, 5], np.int32)
    pt = np.array([1, 0, 3, 4, 7, 6], np.int32)
    m = {"crosscheck": _lib.SLAM_MATCH_CROSSCHECK, "knn2": _lib.SLAM_MATCH_KNN2, "mutual": _lib.SLAM_MATCH_CROSSCHECK_MUTUAL}[mode]
    d_desc, d_count = dev(desc), dev(counts)
    small = ops.match_keys_batched(d_desc, d_count, dev(pq), dev(pt), mode=m)
    for k in range(len(pq)):                                                  # one problem per call: B split the most
        alone = ops.match_keys_batched(d_desc, d_count, dev(pq[k:k + 1]), dev(pt[k:k + 1]), mode=m)
        for a, b in zip(small, alone):
            if a is not None:
                assert np.array_equal(a.cpu().numpy()[k], b.cpu().numpy()[0])
    if mode != "knn2":
        idx, dist = (x.cpu().numpy() for x in ops.unpack_keys(small[0]))
        for k in range(len(pq)):
            nq, nt = counts[pq[k]], counts[pt[k]]
            wq, wt, wd = c_oracle.match_crosscheck(desc[pq[k], :nq, :61], desc[pt[k], :nt, :61], orc.NORM_HAMMING2, mutual=mode == "mutual")
            got_q = np.nonzero(idx[k] >= 0)[0]
            assert np.array_equal(got_q, wq) and np.array_equal(idx[k][got_q], wt) and np.array_equal(dist[k][got_q], wd)
    big = ops.match_keys_batched(d_desc, d_count, dev(np.tile(pq, 70)), dev(np.tile(pt, 70)), mode=m)
    for a, b in zip(small, big):
        if a is None:
            assert b is None
            continue
        a, b = a.cpu().numpy(), b.cpu().numpy().reshape(70, len(pq), n_max)
        assert (b == a[None]).all()
    for k in range(len(pq)):
        nq, nt = counts[pq[k]], counts[pt[k]]
        if nq == 0:
            continue
        one = ops.match_keys(dev(desc[pq[k], :nq]), dev(desc[pt[k], :nt]), orc.NORM_HAMMING2, m)
        for a, b in zip(small, one):
            if a is not None:
                assert np.array_equal(a.cpu().numpy()[k, :nq], b.cpu().numpy())


def test_batched_knn_with_ratio_test_on_the_device():
    """match(..., REGULAR_KNN) for a whole batch (utills.py:112-119): two nearest rows + significance_test
    `d1 < RATIO * d2` in float64 applied by the resolve kernel; held to the oracle incl. exact ties of the comparison."""
    q, t, plan = _ratio_case()
    rng = np.random.default_rng(1)
    imgs = [q, t, rng.integers(0, 8, (100, 61), dtype=np.uint8), rng.integers(0, 8, (120, 61), dtype=np.uint8),
            rng.integers(0, 8, (1, 61), dtype=np.uint8)]
    n_max = (max(len(i) for i in imgs) + 63) // 64 * 64
    desc = np.zeros((len(imgs), n_max, 64), np.uint8)
    counts = np.array([len(i) for i in imgs], np.int32)
    for k, im in enumerate(imgs):
        desc[k, :len(im), :61] = im
    pq, pt = np.array([0, 2, 3, 2], np.int32), np.array([1, 3, 2, 4], np.int32)          # last: a single train row
    for ratio in (0.6, 0.8, 1.0):
        k1, k2 = ops.match_keys_batched(dev(desc), dev(counts), dev(pq), dev(pt), mode=_lib.SLAM_MATCH_KNN2, ratio=ratio)
        i1, d1 = (x.cpu().numpy() for x in ops.unpack_keys(k1))
        for k in range(len(pq)):
            a, b = imgs[pq[k]], imgs[pt[k]]
            got_q = np.nonzero(i1[k] >= 0)[0]
            if len(b) < 2:
                assert len(got_q) == 0                     # no second neighbour: nothing can pass (the reference cannot unpack it)
                continue
            wq, wt, wd = orc.knn_ratio_match(a, b, orc.NORM_HAMMING2, ratio)
            assert np.array_equal(got_q, wq) and np.array_equal(i1[k][got_q], wt) and np.array_equal(d1[k][got_q], wd)
        assert 0 < (i1[0] >= 0).sum() < len(q)


def test_batched_match_and_join_bit_exact():
    seq, desc, xy, count = _small_sequence()
    F, _, n_max, _ = desc.shape
    d_desc = dev(desc.reshape(2 * F, n_max, 64))
    d_xy = dev(xy.reshape(2 * F, n_max, 2))
    d_count = dev(count.reshape(2 * F))
    fr = torch.arange(F, dtype=torch.int32).cuda()
    skey, _ = ops.match_keys_batched(d_desc, d_count, (2 * fr).int(), (2 * fr + 1).int())
    f0 = torch.arange(F - 1, dtype=torch.int32).cuda()
    f1 = f0 + 1
    tkey, _ = ops.match_keys_batched(d_desc, d_count, (2 * f0).int(), (2 * f1).int())
    tcount, kp, xy0, xy1 = ops.rect_filter_join(d_xy, d_count, skey, tkey, f0, f1, 2.0)
    sidx, sdist = (x.cpu().numpy() for x in ops.unpack_keys(skey))
    tcount, kp, xy0, xy1 = tcount.cpu().numpy(), kp.cpu().numpy(), xy0.cpu().numpy(), xy1.cpu().numpy()

    def frame(i):
        n0, n1 = count[i]
        return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])

    stereo = []
    for i in range(F):
        f = frame(i)
        q, t, d = c_oracle.match_crosscheck(f["dsc_l"], f["dsc_r"])
        qi = np.nonzero(sidx[i] >= 0)[0]
        assert np.array_equal(qi, q) and np.array_equal(sidx[i][qi], t) and np.array_equal(sdist[i][qi], d)
        rec, _ = orc.rect_filter(f["xy_l"][:, 1], f["xy_r"][:, 1], q, t)
        stereo.append((q, t, d, rec))
    for p in range(F - 1):
        a, b = frame(p), frame(p + 1)
        tq, tt, _ = c_oracle.match_crosscheck(a["dsc_l"], b["dsc_l"])
        ia, ib, _ = orc.join_4view(stereo[p][0], stereo[p][3], stereo[p + 1][0], stereo[p + 1][3], tq, tt)
        want = np.stack([stereo[p][0][ia], stereo[p][1][ia], stereo[p + 1][0][ib], stereo[p + 1][1][ib]], 1)
        T = tcount[p]
        assert T == len(want) and T > 50
        assert np.array_equal(kp[p, :T], want)
        assert np.array_equal(xy0[p, :T, :2], a["xy_l"][want[:, 0]]) and np.array_equal(xy0[p, :T, 2:], a["xy_r"][want[:, 1]])
        assert np.array_equal(xy1[p, :T, :2], b["xy_l"][want[:, 2]]) and np.array_equal(xy1[p, :T, 2:], b["xy_r"][want[:, 3]])


def _rel_err(a, b):
    return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-300)


def test_triangulate_against_reference_golden(golden):
    xl, xr, want = golden["tri_xl"], golden["tri_xr"], golden["tri_X"]
    got = ops.triangulate_f64(K @ M1, K @ M2, dev(xl), dev(xr)).cpu().numpy()
    err = _rel_err(got, want)
    assert np.all(err <= REL_TOL), err.max()            # north_star: 1e-5 relative
    # float32-coordinate batched entry point, ragged segments
    xy = np.zeros((2, 448, 4), np.float32)
    xy[0, :400] = np.hstack([xl, xr]).astype(np.float32)
    xy[1, :100] = xy[0, 100:200]
    cnt = np.array([400, 100], np.int32)
    X = ops.triangulate(K @ M1, K @ M2, dev(xy), dev(cnt)).cpu().numpy()
    assert np.all(_rel_err(X[0, :400], want) <= REL_TOL) and np.all(_rel_err(X[1, :100], want[100:200]) <= REL_TOL)
    assert not X[1, 100:].any() and not X[0, 400:].any()


def test_consensus_masks_bit_exact_against_reference_golden(golden):
    g = golden
    X, l1, r1 = g["cons_X"], g["cons_l1"], g["cons_r1"]
    m = ops.consensus_match_f64(K, M2, g["cons_T"], dev(X), dev(l1), dev(r1), 4.0).cpu().numpy().astype(bool)
    assert np.array_equal(m, g["cons_mask"])
    Tnan = np.hstack([np.eye(3), np.full((3, 1), np.nan)])
    assert not ops.consensus_match_f64(K, M2, Tnan, dev(X), dev(l1), dev(r1), 4.0).cpu().numpy().any()
    # batched scoring entry point: counts and bit masks for 48 hypotheses (+1 failed, +1 NaN)
    n, stride = len(X), 640
    Xs = np.zeros((1, stride, 3))
    Xs[0, :n] = X
    xy1 = np.zeros((1, stride, 4), np.float32)
    xy1[0, :n] = np.hstack([l1, r1]).astype(np.float32)                 # golden observations are float32-exact
    pose = np.concatenate([g["score_T"], g["score_T"][:1], Tnan[None]])[None]
    status = np.ones((1, pose.shape[1]), np.uint8)
    status[0, -2] = 0
    cnt, mask = ops.score_hypotheses(K, M2, dev(Xs), dev(xy1), dev(np.array([n], np.int32)), dev(pose), dev(status), 4.0,
                                     want_mask=True)
    cnt, mask = cnt.cpu().numpy()[0], ops.mask_words_to_bool(mask.cpu().numpy()[0], n)
    H = len(g["score_T"])
    assert np.array_equal(mask[:H], g["score_masks"])
    assert np.array_equal(cnt[:H], g["score_masks"].sum(1))
    assert cnt[H] == 0 and cnt[H + 1] == 0 and not mask[H:].any()


def _general_rig():
    """A calibration with skew and a right camera that is rotated and shifted in all axes: none of the structure the
    rectified-rig scoring kernel relies on."""
    Kg = np.array([[700.0, 1.5, 600.0], [0.0, 710.0, 180.0], [0.0, 0.0, 1.0]])
    M2g = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.002, -0.004, 0.001])), np.array([[-0.54], [0.003], [0.002]])])
    return Kg, M2g


@pytest.mark.parametrize("rig", ["kitti", "general"])
def test_scoring_bit_exact_large_random(rig):
    K, M2 = (orc.K, orc.M2) if rig == "kitti" else _general_rig()
    rng = np.random.default_rng(77)
    n, H, stride = 1900, 96, 1920
    X = np.stack([rng.uniform(-30, 30, n), rng.uniform(-5, 4, n), rng.uniform(-5, 100, n)], 1)
    T0 = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.004, -0.02, 0.003])), np.array([[0.03], [-0.01], [-0.9]])])
    l1 = (orc.project(X, K @ T0) + rng.normal(0, 2.5, (n, 2))).astype(np.float32)
    r1 = (orc.project(X, K @ orc.compose(T0, M2)) + rng.normal(0, 2.5, (n, 2))).astype(np.float32)
    Ts = np.array([np.hstack([orc.rodrigues_vec_to_mat(np.array([0.004, -0.02, 0.003]) + rng.normal(0, 0.002, 3)),
                              (T0[:, 3] + rng.normal(0, 0.02, 3))[:, None]]) for _ in range(H)])
    # poses with a non-finite entry in each row in turn, a signed zero, and a point of zero depth for pose 0
    Ts[1, 0, 1], Ts[2, 1, 3], Ts[3, 2, 0], Ts[4, 0, 3], Ts[5, 1, 0] = np.nan, np.inf, -np.inf, np.nan, -0.0
    X[7] = -np.linalg.solve(Ts[0, :, :3], Ts[0, :, 3]) + Ts[0, :, :3].T @ np.array([0.3, 0.1, 0.0])
    with np.errstate(all="ignore"):
        want = np.array([orc.consensus_match(K, Ts[h], M2, X, l1.astype(np.float64), r1.astype(np.float64), 2) for h in range(H)])
    Xs = np.zeros((1, stride, 3))
    Xs[0, :n] = X
    xy1 = np.zeros((1, stride, 4), np.float32)
    xy1[0, :n] = np.hstack([l1, r1])
    cnt, mask = ops.score_hypotheses(K, M2, dev(Xs), dev(xy1), dev(np.array([n], np.int32)), dev(Ts[None]),
                                     dev(np.ones((1, H), np.uint8)), 4.0, want_mask=True)
    got = ops.mask_words_to_bool(mask.cpu().numpy()[0], n)
    assert np.array_equal(got, want)
    assert np.array_equal(cnt.cpu().numpy()[0], want.sum(1))
    assert 0.05 < want.mean() < 0.95


@pytest.mark.parametrize("rig", ["kitti", "general"])
@pytest.mark.parametrize("xy_dtype", [np.float32, np.float64])
def test_scoring_many_hypotheses_per_segment_lane_per_hypothesis_kernel(rig, xy_dtype):
    """From 256 hypotheses per segment the counting path runs one LANE per hypothesis (score_lanes_kernel: the later
    RANSAC rounds).  Counts against the oracle and against the wavefront-per-hypothesis kernel (the mask path), on
    segments of 0 ... 700 points (below, at and above one 256-point LDS tile), with failed solves, NaN and infinite
    poses, a point on the camera plane, and a hypothesis count that does not fill the last workgroup."""
    K, M2 = (orc.K, orc.M2) if rig == "kitti" else _general_rig()
    rng = np.random.default_rng(91)
    sizes = [0, 1, 5, 20, 63, 64, 255, 256, 257, 513, 700]
    S, H, stride = len(sizes), 300, 704
    T0 = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.004, -0.02, 0.003])), np.array([[0.03], [-0.01], [-0.9]])])
    Xs = np.zeros((S, stride, 3))
    xy1 = np.zeros((S, stride, 4), xy_dtype)
    Ts = np.zeros((S, H, 3, 4))
    status = (rng.random((S, H)) > 0.1).astype(np.uint8)
    want = np.zeros((S, H), np.int64)
    for s, n in enumerate(sizes):
        X = np.stack([rng.uniform(-30, 30, n), rng.uniform(-5, 4, n), rng.uniform(-5, 100, n)], 1)
        l1 = (orc.project(X, K @ T0) + rng.normal(0, 2.5, (n, 2))).astype(np.float32)
        r1 = (orc.project(X, K @ orc.compose(T0, M2)) + rng.normal(0, 2.5, (n, 2))).astype(np.float32)
        for h in range(H):
            # a third of the hypotheses close to the truth (points near the 4 px radius), a third moderately off (some
            # blocks of points certainly outside, some not), a third garbage (what an outlier in the sample produces)
            sr, st_ = ((0.002, 0.02), (0.02, 0.3), (0.8, 5.0))[h % 3]
            Ts[s, h] = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.004, -0.02, 0.003]) + rng.normal(0, sr, 3)),
                                  (T0[:, 3] + rng.normal(0, st_, 3))[:, None]])
        Ts[s, 1, 0, 1], Ts[s, 2, 1, 3], Ts[s, 3, 2, 0], Ts[s, 4, 0, 3], Ts[s, 5, 1, 0] = np.nan, np.inf, -np.inf, np.nan, -0.0
        Ts[s, 6] = np.nan                                              # what a degenerate sample produces
        Ts[s, 299, 2, 3] = np.inf
        status[s, :8] = 1
        if n > 7:
            X[7] = -np.linalg.solve(Ts[s, 0, :, :3], Ts[s, 0, :, 3]) + Ts[s, 0, :, :3].T @ np.array([0.3, 0.1, 0.0])
        Xs[s, :n], xy1[s, :n] = X, np.hstack([l1, r1])
        with np.errstate(all="ignore"):
            for h in range(H):
                if status[s, h] and n:
                    want[s, h] = orc.consensus_match(K, Ts[s, h], M2, X, l1.astype(np.float64), r1.astype(np.float64), 2).sum()
    args = (K, M2, dev(Xs), dev(xy1), dev(np.array(sizes, np.int32)), dev(Ts), dev(status), 4.0)
    cnt, _ = ops.score_hypotheses(*args)                               # >= 256 hypotheses, no mask: lane per hypothesis
    cnt_w, _ = ops.score_hypotheses(*args, want_mask=True)             # mask path: wavefront per hypothesis
    assert np.array_equal(cnt.cpu().numpy(), want)
    assert np.array_equal(cnt_w.cpu().numpy(), want)
    assert want[-1].max() > 100 and (want[:, 6] == 0).all()
    assert (want[-1, 2::3] == 0).mean() > 0.9 and (want[-1, 1::3] > 0).any()      # garbage / partly-off hypotheses are there


@pytest.mark.parametrize("rig", ["unit_rectified", "general"])
def test_scoring_threshold_edge_cases_of_the_division_free_classification(rig):
    """The fast consensus test compares (hx - u hz)^2 + (hy - v hz)^2 with thr^2 hz^2 (1 -+ 1e-9) and leaves everything
    else to the reference's divide + sqrt arithmetic.  Points built to sit where that hand-over must happen: exactly on
    the radius and a few ulps either side, on the camera plane (hz = 0, with and without hx = hy = 0), depths whose
    square underflows or overflows, negative depth, non-finite points and observations; radius 4, 0 and negative.
    Both counting kernels and the mask kernel against the oracle."""
    if rig == "unit_rectified":
        # hx = x, hy = y, hz = z exactly, in both cameras (zero baseline: a tiny depth then projects to modest pixel
        # coordinates in the right camera too; the reference's own rounding error in hx / hz is eps * |u|, so its
        # decisions are only meaningful, and only reproduced, for observations of image-like magnitude)
        K, M2 = np.eye(3), np.hstack([np.eye(3), np.zeros((3, 1))])
    else:
        K, M2 = _general_rig()
    T = np.hstack([np.eye(3), np.zeros((3, 1))])
    u, v = 10.0, 5.0
    pts = []          # (X, left obs, right obs)

    def add(x, y, z, ul=u, vl=v, ur=None, vr=v):
        pts.append(((x, y, z), (ul, vl), ((x / z if z != 0.0 and np.isfinite(z) and np.isfinite(x) else 0.0) if ur is None else ur, vr)))

    for d in (0.0, 2.0 ** -50, -2.0 ** -50, 2.0 ** -44, -2.0 ** -44, 1e-8, -1e-8, 1e-3, -1e-3):
        add(28.0 + d, 10.0, 2.0)                       # left distance 4 + d / 2: on the radius, ulps and 1e-9 either side
        add(20.0, 10.0, 2.0, ur=14.0 + d, vr=5.0)      # same in the right camera: 20 / 2 = 10, 4 px from 14
    add(20.0, 10.0, 2.0)                               # dead centre: distance 0 (the one supporter at radius 0)
    add(3.0, 1.0, 0.0)                                 # on the camera plane: inf
    add(0.0, 0.0, 0.0)                                 # 0 / 0
    add(12e-170, 5e-170, 1e-170)                       # hz^2 underflows; projects to (12, 5): inside by 2 px in both cameras
    add(12e-170, 5e-170, 1e-170, ur=17.0)              # ... right camera 5 px off
    add(12e-140, 5e-140, 1e-140)
    add(12e200, 5e200, 1e200)                          # hz^2 overflows; inside both
    add(14.1e200, 5e200, 1e200)                        # ... outside by 0.1 px
    add(-24.0, -10.0, -2.0)                            # negative depth projects to (12, 5): the reference counts it
    add(np.inf, 1.0, 2.0)
    add(1.0, np.nan, 2.0)
    add(24.0, 10.0, np.inf)
    add(24.0, 10.0, 2.0, ul=np.nan)
    add(24.0, 10.0, 2.0, ur=np.inf)
    add(24.0, 10.0, 2.0, vr=np.nan)
    n, stride, H = len(pts), 64, 320
    X = np.array([p[0] for p in pts])
    l1, r1 = np.array([p[1] for p in pts]), np.array([p[2] for p in pts])
    Xs = np.zeros((1, stride, 3))
    Xs[0, :n] = X
    xy1 = np.zeros((1, stride, 4))
    xy1[0, :n] = np.hstack([l1, r1])
    Ts = np.repeat(T[None, None], H, 1)
    Ts[0, 1::2, 0, 3] = 2.0 ** -45                     # every other hypothesis: shifted by ulps, so hx differs in the last bits
    status = np.ones((1, H), np.uint8)
    for acc, thr in ((2, 4.0), (0, 0.0), (None, -1.0)):
        with np.errstate(all="ignore"):
            want = np.zeros((H, n), bool) if acc is None else np.array([orc.consensus_match(K, Ts[0, h], M2, X, l1, r1, acc) for h in range(H)])
        args = (K, M2, dev(Xs), dev(xy1), dev(np.array([n], np.int32)), dev(Ts), dev(status), thr)
        cnt, _ = ops.score_hypotheses(*args)                                  # lane per hypothesis
        cnt_w, mask = ops.score_hypotheses(*args, want_mask=True)             # wavefront per hypothesis, masks
        cnt2, _ = ops.score_hypotheses(K, M2, dev(Xs), dev(xy1), dev(np.array([n], np.int32)), dev(Ts[:, :40]), dev(status[:, :40]), thr)
        assert np.array_equal(ops.mask_words_to_bool(mask.cpu().numpy()[0], n), want), thr
        assert np.array_equal(cnt.cpu().numpy()[0], want.sum(1)) and np.array_equal(cnt_w.cpu().numpy()[0], want.sum(1)), thr
        assert np.array_equal(cnt2.cpu().numpy()[0], want.sum(1)[:40]), thr
        if rig == "unit_rectified" and acc == 2:
            # on the radius counts (the reference's test is <=); 2^-45 px outside does not; 2^-45 px inside does
            assert want[0, 0] and want[0, 1] and not want[0, 6] and not want[0, 7] and want[0, 8] and want[0, 9]
            assert want[0, 18] and not want[0, 19] and not want[0, 20] and want[0, 21] and not want[0, 22] and want[0, 23]
            assert want[0, 24] and not want[0, 25] and want[0, 26] and not want[0, 27:].any()
        if rig == "unit_rectified" and acc == 0:
            assert want[0].sum() >= 1


def test_device_sample_stream_equals_numpy():
    """slam_randint_draws against np.random.seed / np.random.randint itself: bounds at and around powers of two (the
    rejection rate's extremes), bound 1 (numpy consumes nothing), ranges that start mid-stream and straddle the 624-word
    generator blocks, 32-bit seeds."""
    cases = [(0, 2, 0), (1000, 1400, 0), (1234, 1025, 64), (99, 1024, 37), (7, 3, 500), (2 ** 32 - 1, 65537, 7), (5, 1, 3),
             (31337, 1196, 1000), (42, 700, 2999), (17, 2047, 155), (18, 2048, 156), (19, 2049, 157)]
    n_draw = 96
    seed = np.array([c[0] for c in cases], np.uint32)
    count = np.array([c[1] for c in cases], np.int32)
    begin = np.array([c[2] for c in cases], np.int32)
    got = ops.randint_draws(dev(seed.view(np.int32)), dev(count), dev(begin), n_draw).cpu().numpy()
    for r, (s, n, b) in enumerate(cases):
        np.random.seed(s)
        want = np.stack([np.random.randint(0, n, 4) for _ in range(b + n_draw)])[b:]
        assert np.array_equal(got[r], want), (s, n, b)


def _ransac_problem(rng, n=500, frac_out=0.2):
    T0 = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.01, -0.03, 0.005])), np.array([[0.05], [-0.02], [-0.8]])])
    X = np.stack([rng.uniform(-20, 20, n), rng.uniform(-4, 3, n), rng.uniform(4, 70, n)], 1)
    l = orc.project(X, K @ T0) + rng.normal(0, 0.3, (n, 2))
    r = orc.project(X, K @ orc.compose(T0, M2)) + rng.normal(0, 0.3, (n, 2))
    bad = rng.random(n) < frac_out
    l[bad] += rng.normal(0, 30, (int(bad.sum()), 2))
    return T0, X, l.astype(np.float32), r.astype(np.float32)


def test_p3p_hypotheses_against_oracle():
    rng = np.random.default_rng(5)
    T0, X, l, r = _ransac_problem(rng)
    n, stride, H = len(X), 512, 400
    sample = rng.integers(0, n, (1, H, 4)).astype(np.int32)
    sample[0, 0] = [3, 3, 9, 11]            # repeated point: degenerate, NaN pose but status 1
    sample[0, 1] = [5, 6, 6, 6]
    Xs = np.zeros((1, stride, 3))
    Xs[0, :n] = X
    xy1 = np.zeros((1, stride, 4), np.float32)
    xy1[0, :n] = np.hstack([l, r])
    pose, status = ops.p3p_hypotheses(K, dev(Xs), dev(xy1), dev(np.array([n], np.int32)), dev(sample))
    pose, status = pose.cpu().numpy()[0], status.cpu().numpy()[0]
    n_ok = n_fail = n_nan = 0
    with np.errstate(all="ignore"):
        for h in range(H):
            idx = sample[0, h]
            want = orc.pnp(X[idx], l[idx].astype(np.float64), K, orc.SOLVEPNP_AP3P)
            if want is None:
                assert status[h] == 0
                n_fail += 1
                continue
            assert status[h] == 1
            if not np.all(np.isfinite(want)):
                assert not np.all(np.isfinite(pose[h]))
                n_nan += 1
                continue
            # 1e-5 relative on the pose (north_star); ill-conditioned minimal samples amplify rounding, so the
            # norm is taken over the whole 3x4 matrix
            err = np.linalg.norm(pose[h] - want) / np.linalg.norm(want)
            assert err <= REL_TOL, (h, err)
            n_ok += 1
    assert n_ok > 200 and n_nan >= 2


def test_refit_against_oracle():
    rng = np.random.default_rng(8)
    T0, X, l, r = _ransac_problem(rng, n=700, frac_out=0.25)
    n, stride = len(X), 704
    with np.errstate(all="ignore"):
        Th = orc.pnp(X[[1, 2, 3, 4]], orc.project(X[[1, 2, 3, 4]], K @ T0), K, orc.SOLVEPNP_AP3P)
        mask = orc.consensus_match(K, Th, M2, X, l.astype(np.float64), r.astype(np.float64), 2)
        want = orc.pnp(X[mask], l[mask].astype(np.float64), K, orc.SOLVEPNP_ITERATIVE)
    Xs = np.zeros((2, stride, 3))
    Xs[:, :n] = X
    xy1 = np.zeros((2, stride, 4), np.float32)
    xy1[:, :n] = np.hstack([l, r])
    few = np.zeros(n, bool)
    few[:5] = True                                                   # < 6 supporters -> pose None
    words = ops.bool_to_mask_words(np.stack([mask, few]), stride)
    pose, status = ops.refit_pnp(K, dev(Xs), dev(xy1), dev(np.array([n, n], np.int32)), dev(words), dev(np.stack([Th, Th])))
    pose, status = pose.cpu().numpy(), status.cpu().numpy()
    assert list(status) == [1, 0] and np.isnan(pose[1]).all()
    err = np.linalg.norm(pose[0] - want) / np.linalg.norm(want)
    assert err <= REL_TOL, err                                        # north_star: 1e-5 relative


def _reproj_cost(T, X, l):
    return float(np.sum((orc.project(X, K @ T) - l) ** 2))


@pytest.mark.parametrize("case", ["near_planar", "low_parallax", "coincident_points"])
def test_refit_on_ill_conditioned_supporter_sets(case):
    """Refit parity is pinned to the oracle's minimiser only (cv2's SOLVEPNP_ITERATIVE is absent offline, DESIGN.md
    section 2), and the two solvers start differently: the kernel from the winning RANSAC pose, the oracle / cv2 from a DLT
    initialisation.  So on ill-conditioned sets the basin is NOT assumed shared: the kernel's pose must be a minimum
    at least as good as the oracle's (cost no higher beyond rounding), equal to it within 1e-5 where the costs agree,
    with status 1; supporters that are all one point leave the normal equations singular: the pose then stays finite
    (the RANSAC winner, possibly moved along the null space without raising the cost), status 1."""
    rng = np.random.default_rng({"near_planar": 1, "low_parallax": 2, "coincident_points": 3}[case])
    T0 = np.hstack([orc.rodrigues_vec_to_mat(np.array([0.01, -0.03, 0.005])), np.array([[0.05], [-0.02], [-0.8]])])
    n = 400
    if case == "near_planar":
        X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 2, n), 20.0 + rng.normal(0, 1e-3, n)], 1)   # a wall, 1 mm rough
    elif case == "low_parallax":
        X = np.stack([rng.uniform(-300, 300, n), rng.uniform(-60, 40, n), rng.uniform(900, 1000, n)], 1)  # everything ~1 km away
    else:
        X = np.tile(np.array([[1.0, -0.5, 12.0]]), (n, 1))
    l = (orc.project(X, K @ T0) + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    r = (orc.project(X, K @ orc.compose(T0, M2)) + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    stride = 448
    Xs = np.zeros((1, stride, 3)); Xs[0, :n] = X
    xy1 = np.zeros((1, stride, 4), np.float32); xy1[0, :n] = np.hstack([l, r])
    mask = np.ones((1, n), bool)
    start = T0.copy()
    start[:, 3] += [0.02, -0.01, 0.05]                         # a RANSAC-quality start, a few pixels off
    pose, status = ops.refit_pnp(K, dev(Xs), dev(xy1), dev(np.array([n], np.int32)), dev(ops.bool_to_mask_words(mask, stride)),
                                 dev(start[None]))
    pose, status = pose.cpu().numpy()[0], status.cpu().numpy()
    assert status[0] == 1 and np.isfinite(pose).all()
    c_start, c_got = _reproj_cost(start, X, l.astype(np.float64)), _reproj_cost(pose, X, l.astype(np.float64))
    assert c_got <= c_start * (1 + 1e-12)                      # LM never accepts a worse pose
    if case == "coincident_points":
        R = pose[:, :3]
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-9        # still a rigid pose
        return
    with np.errstate(all="ignore"):
        want = orc.pnp(X, l.astype(np.float64), K, orc.SOLVEPNP_ITERATIVE)
    assert want is not None
    c_want = _reproj_cost(want, X, l.astype(np.float64))
    assert c_got <= c_want * (1 + 1e-9), (c_got, c_want)       # a minimum at least as good as the oracle's
    if abs(c_got - c_want) <= 1e-9 * c_want:
        # same cost: on a flat valley (low parallax) the poses may still sit apart along it, so the pose check is
        # made only where the problem is well posed enough for the oracle itself to be converged
        err = np.linalg.norm(pose - want) / np.linalg.norm(want)
        assert err <= (REL_TOL if case == "near_planar" else 1e-3), err


def _draws_with_repeated_fourth_index(seed, n, draws, slack=8):
    """Draws of the pair's stream whose fourth index repeats one of the first three (np.random.randint draws with
    replacement, utils/utills.py:502): AP3P then picks its root by the reprojection error of a point the roots were
    computed from, i.e. on rounding noise -- in the oracle, in cv2 and here alike -- so the loop's
    counters may differ on such a pair, as in tests/test_gpu_end_to_end.py."""
    np.random.seed(int(seed))
    return sum(int(i[3] in i[:3]) for i in (np.random.randint(0, n, 4) for _ in range(draws + slack)))


def test_full_pipeline_against_oracle_sequence():
    from slam_amd.pipeline import ShardTracker, consecutive_pairs
    seq, desc, xy, count = _small_sequence(n_frames=14, n_features=320, seed=9)
    F, _, n_max, _ = desc.shape
    trk = ShardTracker(F, F - 1, n_max)
    f0, f1 = consecutive_pairs(F)
    seeds = 1000 + np.arange(F - 1)
    res = trk.run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)), f0, f1, seeds)
    kp = res.tracked_kp.cpu().numpy()

    def frame(i):
        n0, n1 = count[i]
        return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])

    clean = 0
    for p in range(F - 1):
        np.random.seed(int(seeds[p]))
        st = {}
        with np.errstate(all="ignore"):
            want = orc.track_pair(frame(p), frame(p + 1), stats=st)
        T = len(want["kp"])
        assert res.stats[p, 0] == T
        assert np.array_equal(kp[p, :T], want["kp"])
        assert not res.mask[p, T:].any()
        gt = seq.gt_relative(p + 1)
        assert np.abs(res.pose[p] - gt).max() < 0.05
        if _draws_with_repeated_fourth_index(seeds[p], T, max(st["draws"], int(res.stats[p, 3]))):
            continue                              # a draw decided on rounding noise: tracked set and pose quality only
        clean += 1
        assert res.stats[p, 2] == st["iterations"] and res.stats[p, 3] == st["draws"] and res.stats[p, 4] == st["best_draw"]
        assert np.array_equal(res.mask[p, :T], want["mask"])
        assert res.stats[p, 1] == want["mask"].sum()
        err = np.linalg.norm(res.pose[p] - want["T"]) / np.linalg.norm(want["T"])
        assert err <= REL_TOL, (p, err)
    assert clean >= 4, clean


@pytest.mark.parametrize("how", ["mutual", "knn_ratio"])
def test_full_pipeline_other_match_methods_against_oracle(how):
    """The shard pipeline matches the way match() does (utills.py:105-119): with the mutual cross-check rule, and with
    REGULAR_KNN (two nearest + ratio 0.6, no cross-check: several left0 key-points may land on one left1 key-point).
    Tracked sets bit-exact against the oracle run with the same matcher; masks and counts equal; poses within REL_TOL."""
    from slam_amd.config import MUTUAL, REGULAR_KNN, HotPathConfig
    from slam_amd.pipeline import ShardTracker, consecutive_pairs
    seq, desc, xy, count = _small_sequence(n_frames=14, n_features=320, seed=9)
    F, _, n_max, _ = desc.shape
    clean = 0
    cfg = HotPathConfig(crosscheck_rule=MUTUAL) if how == "mutual" else HotPathConfig(match_method=REGULAR_KNN)
    matcher = orc.bf_match_mutual if how == "mutual" else orc.knn_ratio_match
    f0, f1 = consecutive_pairs(F)
    seeds = 1000 + np.arange(F - 1)
    args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)), f0, f1, seeds)
    res = ShardTracker(F, F - 1, n_max, cfg).run(*args)
    base = ShardTracker(F, F - 1, n_max).run(*args)
    kp = res.tracked_kp.cpu().numpy()

    def frame(i):
        n0, n1 = count[i]
        return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])

    for p in range(F - 1):
        np.random.seed(int(seeds[p]))
        st = {}
        with np.errstate(all="ignore"):
            want = orc.track_pair(frame(p), frame(p + 1), stats=st, matcher=matcher)
        T = len(want["kp"])
        assert res.stats[p, 0] == T and T > 50
        assert np.array_equal(kp[p, :T], want["kp"]) and not res.mask[p, T:].any()
        if _draws_with_repeated_fourth_index(seeds[p], T, max(st["draws"], int(res.stats[p, 3]))):
            continue                              # a draw decided on rounding noise (see the helper)
        clean += 1
        assert res.stats[p, 2] == st["iterations"] and res.stats[p, 3] == st["draws"] and res.stats[p, 4] == st["best_draw"]
        assert np.array_equal(res.mask[p, :T], want["mask"])
        err = np.linalg.norm(res.pose[p] - want["T"]) / np.linalg.norm(want["T"])
        assert err <= REL_TOL, (p, err)
    assert clean >= 4, clean
    # the method really changes the match lists on this data (stats columns 6, 7: stereo / temporal matches per pair;
    # the mutual rule keeps a subset of the legacy rule's pairs, most of the others die in the 4-view join anyway)
    assert not np.array_equal(res.stats[:, 6:8], base.stats[:, 6:8])
    if how == "mutual":
        assert (res.stats[:, 6:8] <= base.stats[:, 6:8]).all()


def test_longer_sequence_equals_oracle_wherever_no_draw_repeats_its_fourth_index():
    """End-to-end bar: tracked sets are bit-exact for every pair; winning draw, supporter mask,
    iteration and draw counts are bit-exact for every pair whose sample stream holds no draw with a repeated fourth
    index (AP3P then picks its root on rounding noise, in the oracle and in cv2 alike); poses within REL_TOL there."""
    from slam_amd.pipeline import ShardTracker, consecutive_pairs
    cfg = SynthConfig(n_frames=48, n_features=700, seed=21, ragged=17)
    seq = SynthSequence(cfg)
    F = cfg.n_frames
    desc, xy, count = seq.frames(0, F)
    n_max = cfg.n_max
    f0, f1 = consecutive_pairs(F)
    seeds = 1000 + np.arange(F - 1)
    res = ShardTracker(F, F - 1, n_max, n_chunks=2).run(dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)),
                                                        dev(count.reshape(2 * F)), f0, f1, seeds)
    kp = res.tracked_kp.cpu().numpy()

    def frame(i):
        n0, n1 = count[i]
        return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])

    clean = 0
    for p in range(F - 1):
        np.random.seed(int(seeds[p]))
        st = {}
        with np.errstate(all="ignore"):
            want = orc.track_pair(frame(p), frame(p + 1), stats=st)
        T = len(want["kp"])
        assert res.stats[p, 0] == T and np.array_equal(kp[p, :T], want["kp"]), p
        np.random.seed(int(seeds[p]))
        draws = max(st["draws"], int(res.stats[p, 3]))
        idx = np.stack([np.random.randint(0, T, 4) for _ in range(draws)])
        if any(i[3] in i[:3] for i in idx):
            continue
        clean += 1
        assert (res.stats[p, 2], res.stats[p, 3], res.stats[p, 4]) == (st["iterations"], st["draws"], st["best_draw"]), p
        assert np.array_equal(res.mask[p, :T], want["mask"]), p
        err = np.linalg.norm(res.pose[p] - want["T"]) / np.linalg.norm(want["T"])
        assert err <= REL_TOL, (p, err)
    assert clean >= (F - 1) // 2          # the property must have been exercised


def test_pipeline_with_plain_hamming_norm_and_a_general_rig():
    """The shard pipeline off its defaults: NORM_HAMMING instead of NORM_HAMMING2 (the README of the reference says
    "Hamming", its code NORM_HAMMING2: both must work) and a calibration / right camera without the rectified
    structure the fast scoring variant needs.  Tracked sets bit-exact, masks and counts equal to the oracle's."""
    from slam_amd.config import HotPathConfig, NORM_HAMMING
    from slam_amd.pipeline import ShardTracker, consecutive_pairs
    seq, desc, xy, count = _small_sequence(n_frames=5, n_features=320, seed=12)
    F, _, n_max, _ = desc.shape
    Kg, M2g = _general_rig()
    f0, f1 = consecutive_pairs(F)
    seeds = 500 + np.arange(F - 1)
    args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)), f0, f1, seeds)

    def frame(i):
        n0, n1 = count[i]
        return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])

    for cfg, kw in ((HotPathConfig(norm=NORM_HAMMING), dict(norm=orc.NORM_HAMMING)),
                    (HotPathConfig(K=Kg, M2=M2g), dict(calib=Kg, M2_=M2g))):
        res = ShardTracker(F, F - 1, n_max, cfg).run(*args)
        kp = res.tracked_kp.cpu().numpy()
        for p in range(F - 1):
            np.random.seed(int(seeds[p]))
            st = {}
            with np.errstate(all="ignore"):
                want = orc.track_pair(frame(p), frame(p + 1), stats=st, **kw)
            T = len(want["kp"])
            assert res.stats[p, 0] == T and np.array_equal(kp[p, :T], want["kp"]), p
            idx = np.random.RandomState(int(seeds[p])).randint(0, T, (max(st["draws"], int(res.stats[p, 3])), 4))
            if any(i[3] in i[:3] for i in idx):
                continue            # AP3P root choice on rounding noise
            assert (res.stats[p, 2], res.stats[p, 4]) == (st["iterations"], st["best_draw"]), p
            assert np.array_equal(res.mask[p, :T], want["mask"]), p
            if want["T"] is not None:
                err = np.linalg.norm(res.pose[p] - want["T"]) / np.linalg.norm(want["T"])
                assert err <= REL_TOL, (p, err)


def test_chunked_two_stream_pipeline_equals_single_pass():
    from slam_amd.pipeline import ShardTracker, consecutive_pairs
    seq, desc, xy, count = _small_sequence(n_frames=12, n_features=300, seed=4)
    F, _, n_max, _ = desc.shape
    f0, f1 = consecutive_pairs(F)
    seeds = 777 + np.arange(F - 1)
    args = (dev(desc.reshape(2 * F, n_max, 64)), dev(xy.reshape(2 * F, n_max, 2)), dev(count.reshape(2 * F)), f0, f1, seeds)
    one = ShardTracker(F, F - 1, n_max).run(*args)
    kp_one = one.tracked_kp.cpu().numpy().copy()
    for n_chunks in (2, 3, 5):
        many = ShardTracker(F, F - 1, n_max, n_chunks=n_chunks).run(*args)
        assert np.array_equal(many.stats[:, :5], one.stats[:, :5])
        assert np.array_equal(many.mask, one.mask)
        assert np.array_equal(many.pose, one.pose, equal_nan=True)
        T = one.stats[:, 0]
        kp_many = many.tracked_kp.cpu().numpy()
        for p in range(F - 1):
            assert np.array_equal(kp_many[p, :T[p]], kp_one[p, :T[p]])

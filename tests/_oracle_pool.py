"""CPU oracle runs spread over worker processes (test infrastructure): every worker rebuilds the synthetic sequence from
its configuration (frames are pure functions of (config, frame index)) and runs oracle.track_pair on the pairs it is
handed.  Spawned, not forked: the test process has an initialised HIP runtime."""
import multiprocessing as mp
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.join(_ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

_SEQ = None
_KW = None
_MATCHER = None
_SLACK = 64


def _init(cfg_kwargs, ransac_kw, matcher=None, slack=64):
    global _SEQ, _KW, _MATCHER, _SLACK
    _SLACK = slack
    import c_oracle
    import slam_oracle as orc
    from slam_amd.synth import SynthConfig, SynthSequence
    orc.bf_match = lambda a, b, norm=7: c_oracle.match_crosscheck_w64(c_oracle.pad64(a), c_oracle.pad64(b), norm)
    _SEQ = SynthSequence(SynthConfig(**cfg_kwargs))
    _KW = dict(ransac_kw)
    # what match() does (utils/utills.py:105-119): default cross-check (legacy rule), or the mutual rule / kNN + ratio
    if matcher == "mutual":
        _MATCHER = lambda a, b, norm=7: c_oracle.match_crosscheck(a, b, norm, mutual=True)
    elif matcher == "knn_ratio":
        def _knn(a, b, norm=7):
            idx, dist = c_oracle.knn2(a, b, norm)
            keep = dist[:, 0].astype(np.float64) < orc.RATIO * dist[:, 1].astype(np.float64)
            q = np.nonzero(keep)[0].astype(np.int32)
            return q, idx[q, 0], dist[q, 0]
        _MATCHER = _knn


def _frame(i):
    f = _SEQ.frame(int(i))
    n0, n1 = f["count"]
    return dict(dsc_l=f["desc"][0, :n0, :61], dsc_r=f["desc"][1, :n1, :61], xy_l=f["xy"][0, :n0], xy_r=f["xy"][1, :n1])


def _pair(job):
    """(key, frame0, frame1, seed) -> key, tracked kp, packed mask, iterations, draws, best draw, pose | None, number
    of draws whose fourth index repeats one of the first three (AP3P then picks its root on rounding noise)."""
    import slam_oracle as orc
    key, f0, f1, seed = job
    np.random.seed(int(seed))
    st = {}
    with np.errstate(all="ignore"):
        try:
            w = orc.track_pair(_frame(f0), _frame(f1), stats=st, matcher=_MATCHER, **_KW)
        except ValueError as e:                      # np.random.randint(0, 0, 4): nothing tracked (utils/utills.py:502)
            if str(e) != "low >= high":
                raise
            return key, np.zeros((0, 4), np.int32), np.zeros(0, np.uint8), -1, -1, -1, None, 0
    np.random.seed(int(seed))
    n, noisy = len(w["kp"]), 0
    for _ in range(st["draws"] + _SLACK):            # + slack: the other side may have consumed a few more draws
        i = np.random.randint(0, n, 4)
        noisy += int(i[3] in i[:3])
    return key, w["kp"].astype(np.int32), np.packbits(w["mask"]), st["iterations"], st["draws"], st["best_draw"], w["T"], noisy


def cpu_share():
    """Worker processes for this job: the box's CPU share (OMP_NUM_THREADS is set to it there), at most 16."""
    n = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or len(os.sched_getaffinity(0))
    return max(1, min(n, 16))


def run(cfg_kwargs, jobs, ransac_kw=None, workers=None, matcher=None, slack=64):
    """jobs: iterable of (key, frame0, frame1, seed).  Returns {key: result tuple without the key}.
    matcher: None (cross-check, legacy rule), "mutual" or "knn_ratio".  slack: draws beyond the oracle's own that are
    also examined for a repeated fourth index."""
    workers = workers or cpu_share()
    out = {}
    with mp.get_context("spawn").Pool(workers, initializer=_init, initargs=(cfg_kwargs, ransac_kw or {}, matcher, slack)) as pool:
        for r in pool.imap_unordered(_pair, list(jobs), chunksize=4):
            out[r[0]] = r[1:]
    return out

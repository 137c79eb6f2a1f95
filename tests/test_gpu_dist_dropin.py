"""The reference-shaped whole-movie call under torch.distributed (SURVEY 8 row (e) reached through the reference's own
entry point): two ranks, each tracking its block of frame pairs, must return on EVERY rank what one process returns for
the same global random stream -- poses, inlier percentages, supporters, coordinates, and the track database built
from them.  Both ranks share the box's one GPU and use gloo; on a node it is one GPU per rank and RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from slam_amd import utills as U
from slam_amd.types import DataBase

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_whole_movie_equals_the_single_process_call(golden, tmp_path, world):
    U.set_detection_source(U.ArrayDetections(golden["movie_desc"], golden["movie_xy"], golden["movie_count"]))
    try:
        np.random.seed(7)
        feats, percs, glob, rel = U.find_features_in_consecutive_frames_whole_movie()
        after = np.random.randint(0, 2 ** 31)
        db = DataBase(feats, percs, glob, rel)
    finally:
        U.set_detection_source(None)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SLAM_TEST_BACKEND="gloo", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "_dist_dropin_worker.py"), str(tmp_path)]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-3000:]
    flat, ptr = feats.flat
    t = db.get_table()
    for rank in range(world):
        z = np.load(tmp_path / f"rank{rank}.npz", allow_pickle=False)
        assert np.array_equal(z["flat"], flat) and np.array_equal(z["ptr"], ptr) and len(flat) > 500
        assert z["percs"].tolist() == percs
        assert np.array_equal(z["glob"], np.array(glob)) and np.array_equal(z["rel"], np.array(rel))
        assert np.array_equal(z["xy"], np.asarray(feats.xy_frames))
        assert int(z["n_tracks"]) == db.get_num_tracks() and np.array_equal(z["frame_tracks"], t.frame_tracks)
        assert np.array_equal(z["track_ptr"], t.track_ptr) and np.array_equal(z["obs_xy"], t.obs_xy)
        if rank == 0:
            assert int(z["stream_after"]) == after            # rank 0's global stream advanced exactly as one process's does


@pytest.mark.parametrize("world,plan", [(2, "one"), (3, "one"), (2, "two"), (3, "two")])
def test_sharded_loop_closure_validation_equals_the_single_process_call(golden, tmp_path, world, plan):
    """SURVEY 8(e), the second split north_star names: the flat (candidate, current) pair list of loop-closure
    validation (PoseGraph.py:463-472 -> utils/utills.py:1118-1146) block-partitioned over the ranks.  Every rank must
    return what one process returns for rank 0's random stream: which candidates pass, and every little track (ids,
    frame ids, key-point indices, coordinates); rank 0's global stream must advance as one process's does.  Six pairs
    over 2 and 3 ranks (blocks of 3 and of 2); a current frame without candidates; one frame's candidates alone.
    plan "one": a rank stereo-matches every frame its block touches; plan "two": every frame is stereo-matched on one
    rank, the keys are all-gathered (the path's one exchange step) and the pair blocks run on them."""
    import json
    sys.path.insert(0, HERE)
    import _dist_dropin_worker as W
    U.set_detection_source(U.ArrayDetections(golden["movie_desc"], golden["movie_xy"], golden["movie_count"]))
    want = {}
    try:
        for thr in W.LOOP_CASE["thresholds"]:
            np.random.seed(11)
            out = U.find_loop_cands_by_consensus_batch(W.LOOP_CASE["cands"], W.LOOP_CASE["pgs"], W.LOOP_CASE["curs"], thr)
            want[str(thr)] = dict(summary=W.loop_summary(out), stream_after=int(np.random.randint(0, 2 ** 31)))
        want["one_frame"] = W.loop_summary([U.find_loop_cand_by_consensus([2, 5, 0], [20, 50, 0], 6, 50.0, seeds=np.arange(3) + 5)])
    finally:
        U.set_detection_source(None)
    assert sum(len(c) for c in want["50.0"]["summary"]) == 6 and sum(len(c) for c in want["100.0"]["summary"]) == 0
    assert 0 < sum(len(c) for c in want["96.5"]["summary"]) < 6
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SLAM_TEST_BACKEND="gloo", OMP_NUM_THREADS="2", SLAM_TEST_LOOP_PLAN=plan)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "_dist_dropin_worker.py"), str(tmp_path), "loops"]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-3000:]
    for rank in range(world):
        with open(tmp_path / f"loops_rank{rank}.json") as fh:
            got = json.load(fh)
        for thr in W.LOOP_CASE["thresholds"]:
            assert got[str(thr)]["summary"] == want[str(thr)]["summary"]
            if rank == 0:
                assert got[str(thr)]["stream_after"] == want[str(thr)]["stream_after"]
        assert got["one_frame"] == want["one_frame"]

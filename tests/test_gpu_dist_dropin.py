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

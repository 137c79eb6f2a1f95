"""One rank of tests/test_gpu_dist_dropin.py: the reference-shaped whole-movie call under torch.distributed (started by
torch.distributed.run; both ranks use GPU 0 with the gloo backend on the one-GPU test box, RCCL on a real node)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import utills as U  # noqa: E402
from slam_amd.types import DataBase  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(int(os.environ.get("SLAM_TEST_DEVICE", 0)))
    dist.init_process_group(os.environ.get("SLAM_TEST_BACKEND", "gloo"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"), allow_pickle=False)
    U.set_detection_source(U.ArrayDetections(g["movie_desc"], g["movie_xy"], g["movie_count"]))
    np.random.seed(7 if rank == 0 else 1000 + rank)          # only rank 0's stream may matter
    feats, percs, glob, rel = U.find_features_in_consecutive_frames_whole_movie()
    db = DataBase(feats, percs, glob, rel)
    flat, ptr = feats.flat
    t = db.get_table()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=flat, ptr=ptr, percs=np.array(percs), glob=np.array(glob),
             rel=np.array([np.full((3, 4), np.nan) if r is None else r for r in rel]), xy=np.asarray(feats.xy_frames),
             n_tracks=db.get_num_tracks(), frame_tracks=t.frame_tracks, track_ptr=t.track_ptr, obs_xy=t.obs_xy,
             stream_after=np.random.randint(0, 2 ** 31) if rank == 0 else -1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

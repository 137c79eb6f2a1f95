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


LOOP_CASE = dict(curs=[4, 5, 7, 6], cands=[[3, 0], [1], [], [2, 5, 0]], pgs=[[30, 0], [10], [], [20, 50, 0]],
                 thresholds=(50.0, 96.5, 100.0))


def loop_summary(out):
    """find_loop_cands_by_consensus_batch's result as plain lists (JSON): per current frame, per passing candidate, the
    pose-graph index and every little track's id, frame ids and left / right locations."""
    return [[[int(pg), [[int(t.get_id()), [int(f) for f in t.get_frames_ids()],
                         np.asarray(t.get_left_locations_in_all_frames(), dtype=float).tolist(),
                         np.asarray(t.get_right_locations_in_all_frames(), dtype=float).tolist(),
                         [int(t.get_feature_at_frame(f).get_left_kpt()) for f in t.get_frames_ids()],
                         [int(t.get_feature_at_frame(f).get_right_kpt()) for f in t.get_frames_ids()]] for t in trs]]
             for pg, trs in per_cur] for per_cur in out]


def loops(out_dir, rank):
    """Loop-closure candidate validation with the flat pair list split over the ranks (second split)."""
    import json
    got = {}
    for thr in LOOP_CASE["thresholds"]:
        np.random.seed(11 if rank == 0 else 2000 + rank)     # seeds=None: rank 0 draws them from ITS global stream
        out = U.find_loop_cands_by_consensus_batch(LOOP_CASE["cands"], LOOP_CASE["pgs"], LOOP_CASE["curs"], thr)
        got[str(thr)] = dict(summary=loop_summary(out), stream_after=int(np.random.randint(0, 2 ** 31)) if rank == 0 else -1)
    one = U.find_loop_cand_by_consensus([2, 5, 0], [20, 50, 0], 6, 50.0, seeds=np.arange(3) + 5 if rank == 0 else None)
    got["one_frame"] = loop_summary([one])
    with open(os.path.join(out_dir, f"loops_rank{rank}.json"), "w") as fh:
        json.dump(got, fh)


def main():
    out_dir = sys.argv[1]
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(int(os.environ.get("SLAM_TEST_DEVICE", 0)))
    dist.init_process_group(os.environ.get("SLAM_TEST_BACKEND", "gloo"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"), allow_pickle=False)
    U.set_detection_source(U.ArrayDetections(g["movie_desc"], g["movie_xy"], g["movie_count"]))
    if len(sys.argv) > 2 and sys.argv[2] == "loops":
        U.LOOP_SHARD_PLAN = os.environ.get("SLAM_TEST_LOOP_PLAN", "auto")    # one phase, or stereo keys matched once + all-gathered
        loops(out_dir, rank)
        dist.barrier()
        dist.destroy_process_group()
        return
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

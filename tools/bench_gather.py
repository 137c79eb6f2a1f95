"""Cost of the per-step exchange (slam_amd.dist.gather_shard_results) under RCCL: run under torch.distributed.run with
one or more ranks per GPU box.   python -m torch.distributed.run --nproc-per-node 1 tools/bench_gather.py"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import dist as sdist  # noqa: E402


def main():
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count()))
    dist.init_process_group(os.environ.get("SLAM_BENCH_BACKEND", "nccl"))
    world, rank = dist.get_world_size(), dist.get_rank()
    P = 3449
    a, b = sdist.block_range(P, rank, world)
    pose, stats = np.random.rand(b - a, 3, 4), np.random.randint(0, 1000, (b - a, 8)).astype(np.int32)
    for _ in range(5):
        sdist.gather_shard_results(pose, stats, P)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        sdist.gather_shard_results(pose, stats, P)
    dt = (time.perf_counter() - t0) / n
    if rank == 0:
        print(f"world {world}: gather_shard_results {dt * 1e6:.0f} us per call")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

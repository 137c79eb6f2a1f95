#!/usr/bin/env python
"""Benchmark of the stereo-trajectory hot path (BASELINE.json metric: frame-pairs/s for
match + triangulate + PnP-RANSAC on synthetic KITTI-shaped stereo, 3450 frames x ~2k features per image).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one 3450-frame sequence whose detections are already resident in
HBM, through to relative poses + supporter masks on the host and the composed global trajectory on rank 0.

N = 1 is BASELINE configs[1].  N > 1 is configs[2]: the SAME 3450-frame sequence block-partitioned by frame range over
the N GPUs (strong scaling; the reference's sequential loop utils/utills.py:850 sharded), no data-path collective, one
RCCL all-gather of the relative poses + per-pair counters, composition on rank 0 -- that is `value`.  The weak-scaling
figure (every rank its own 3450-frame range of a longer drive) is measured in the same run and reported beside it
under "weak".  `python bench.py --gpus N` with N > 1 starts the N ranks itself (one process per GPU through
torch.distributed.run); under an external launcher (RANK set) it is one of the ranks.  Rank 0 prints ONE JSON line.

--config stress / loops time BASELINE configs[3] / configs[4] with the same JSON schema; loops also takes --gpus N
(the one candidate-pair list block-partitioned over the ranks, strong scaling).
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Peak used for the matcher's roofline: MI355X vector ALU issue rate for full-rate 32-bit integer ops,
# 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (MI355X_MICROARCH.md: FP32 vector peak 157.3 TFLOP/s = 2 flop x this rate).
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
HBM_PEAK_GBPS = 8000.0
# VALU instructions the matcher executes per descriptor pair (NORM_HAMMING2, 64-byte rows): 8 x (v_xor, v_bitop3,
# v_bcnt) + v_cmp (the key/min update behind it runs for a few percent of the rows); SURVEY section 8(d) counts the
# naive formulation at 80 (+4) per pair.  v_bcnt issues at half rate on gfx950 (tools/valu_peak.hip), so a pair
# costs 8 x (1 + 1 + 2) + 1 = 33 full-rate issue slots.
EXEC_OPS_PER_PAIR = 25
ISSUE_SLOTS_PER_PAIR = 33
# Sustained issue rates measured on this part (tools/valu_peak.hip -> profiles/r1_valu_peak_microbench.txt, 40 launches
# back to back, T lane-ops/s): the fastest full-rate instruction is the ceiling a "fraction of issue slots" is quoted
# against (so that it cannot exceed 1), and the three rates together give the ceiling of the matcher's own mix.
VALU_SUSTAINED_TSLOTS = 66.27     # v_xor_b32
VALU_SUSTAINED_BITOP3, VALU_SUSTAINED_BCNT = 61.09, 36.37
# descriptor pairs/s if every instruction of a pair issued at its own measured sustained rate: 8 xor + 1 cmp, 8 bitop3, 8 bcnt
MIX_CEILING_TPAIRS = 1.0 / (9 / VALU_SUSTAINED_TSLOTS + 8 / VALU_SUSTAINED_BITOP3 + 8 / VALU_SUSTAINED_BCNT)
SURVEY_OPS_PER_PAIR = 80
PMC_SUMMARY = os.path.join("profiles", "r3_final2_pmc_summary.json")
PMC_SUMMARY_FALLBACK = os.path.join("profiles", "r3_pmc_summary.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=("trajectory", "stress", "loops"), default="trajectory",
                    help="trajectory = BASELINE configs[1] (N = 1) / configs[2] (N > 1), the headline; stress = configs[3] "
                         "(8k features, 60 %% outliers, 10 000 hypotheses per pair); loops = configs[4] (3450 frames x 5 candidates)")
    ap.add_argument("--frames", type=int, default=0, help="frames of the sequence (0 = the config's own: 3450; stress 31)")
    ap.add_argument("--features", type=int, default=0, help="features per image (0 = the config's own: 2000; stress 8000)")
    ap.add_argument("--cpu-pairs", type=int, default=192, help="frame pairs timed through the one-thread CPU oracle (rank 0, N=1)")
    ap.add_argument("--cpu-pairs-all-cores", type=int, default=1920, help="frame pairs timed through the oracle on all host cores")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--chunks", type=int, default=0,
                    help="frame-range chunks pipelined over two HIP streams per GPU (0 = the library's plan: 2 for a large shard, 1 for a small one)")
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"), default="auto",
                    help="auto (default): N > 1 times the strong leg (value) and the weak leg (reported beside it); "
                         "weak / strong: only that leg")
    ap.add_argument("--first-chunk", type=int, default=0, help="hypotheses evaluated speculatively per pair in RANSAC round 0 (0 = library default)")
    ap.add_argument("--taper", type=float, default=1.0, help="each chunk's size relative to the previous one (1 = equal)")
    ap.add_argument("--serial-match", type=int, default=-1, choices=(-1, 0, 1),
                    help="1: the chunks' matchers run back to back, each chunk's tail beside the next matcher; 0: chunks on "
                         "different streams start together; -1 (default): the tracker's own choice for the shard size")
    ap.add_argument("--no-ingest", action="store_true", help="skip the PCIe-inclusive leg (reported beside, never as, value)")
    ap.add_argument("--no-dropin", action="store_true", help="skip timing the reference-shaped whole-movie call")
    ap.add_argument("--loops-plan", choices=("auto", "one", "two"), default="auto",
                    help="--config loops with N > 1: one = a rank stereo-matches every frame its pair block touches, no "
                         "data-path collective; two = every frame stereo-matched on one rank, keys all-gathered, then the "
                         "pair blocks; auto = slam_amd.dist.pair_list_plan")
    ap.add_argument("--match-rule", choices=("ocv_legacy", "mutual", "knn"), default="ocv_legacy",
                    help="cross-check rule of the matcher (DESIGN.md section 2), or knn = the reference's REGULAR KNN "
                         "method (two nearest + ratio 0.6, utills.py:112-119); the headline uses the default")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n_gpus: int, argv, port: int):
    """The command `python bench.py --gpus N ...` re-issues itself as when it is not already a rank: one process per
    GPU of this node through torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv) -> int:
    """Parent of an N-rank run.  It never touches the GPU (no HIP call, no torch.cuda initialisation): it starts the
    ranks as a child process group, forwards rank 0's one JSON line and returns the child's exit code."""
    cmd = launch_command(args.gpus, argv, _free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC (RCCL needs it on this pool)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    printed = False
    for line in proc.stdout.decode(errors="replace").splitlines():
        s = line.strip()
        if s.startswith("{") and '"metric"' in s and not printed:
            print(s, flush=True)
            printed = True
        elif s:
            print(s, file=sys.stderr)
    if proc.returncode == 0 and not printed:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return proc.returncode


# ------------------------------------------------------------------------------------------------ one rank
class Rank:
    """Process-group plumbing of one rank (PyTorch is plumbing here: device selection, RCCL, barriers)."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # SLAM_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share
        # devices, collectives run on CPU tensors); the real runs use RCCL, one rank per GPU.
        self.backend = os.environ.get("SLAM_BENCH_BACKEND", "nccl")
        n_dev = torch.cuda.device_count()
        if self.backend == "nccl" and self.world > n_dev:
            raise SystemExit(f"--gpus {args.gpus}: only {n_dev} GPU(s) visible (SLAM_BENCH_BACKEND=gloo rehearses with shared devices)")
        self.local = local % max(1, n_dev)
        torch.cuda.set_device(self.local)
        # Rank 0 prints ONE JSON line: RCCL writes a version banner to stdout when a communicator is created, so file
        # descriptor 1 points at stderr for the run and the line goes to the saved descriptor.
        sys.stdout.flush()
        self.json_fd = os.dup(1)
        os.dup2(2, 1)
        # SLAM_BENCH_FORCE_DIST=1 under a one-process torch.distributed.run: the process group and the gather run at
        # world size 1 too, which rehearses the RCCL calls of the multi-GPU run on a one-GPU box.
        self.use_dist = self.world > 1 or (os.environ.get("SLAM_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
        if self.use_dist:
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local))
            else:
                dist.init_process_group(self.backend)

    def fence(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        if not self.use_dist:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def emit(self, out: dict):
        sys.stdout.flush()
        os.write(self.json_fd, (json.dumps(out) + "\n").encode())

    def close(self):
        if self.use_dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def hot_path_config(args, **kw):
    from slam_amd.config import HotPathConfig
    if args.first_chunk:
        kw["first_chunk"] = args.first_chunk
    if args.match_rule == "mutual":
        kw["crosscheck_rule"] = "mutual"
    if args.match_rule == "knn":
        from slam_amd.config import REGULAR_KNN
        kw["match_method"] = REGULAR_KNN
    return HotPathConfig(**kw)


def trajectory_leg(args, rk: Rank, scaling: str):
    """Data, tracker and the timed loop of one leg.  strong: ONE args.frames-frame movie, rank r holds the frames of
    its block of pairs plus one halo frame (slam_amd.dist.shard_frames).  weak: every rank holds its own movie."""
    torch = rk.torch
    from slam_amd import dist as sdist
    from slam_amd import pipeline
    from slam_amd.synth import SynthConfig, SynthSequence
    F_total = args.frames
    if scaling == "strong":
        pa, pb, fa, fb = sdist.shard_frames(F_total, rk.rank, rk.world)
        scfg = SynthConfig(n_frames=F_total, n_features=args.features, seed=args.seed)
        pairs_total, first_pair = F_total - 1, pa
    else:
        fa, fb = 0, F_total
        scfg = SynthConfig(n_frames=F_total, n_features=args.features, seed=args.seed + 7919 * rk.rank)
        pairs_total, first_pair = rk.world * (F_total - 1), rk.rank * (F_total - 1)
    seq = SynthSequence(scfg)
    t0 = time.time()
    desc, xy, count = seq.frames(fa, fb)                      # outside the timed region
    gen_s = time.time() - t0
    F = fb - fa
    n_max = scfg.n_max
    d_desc = torch.from_numpy(desc.reshape(2 * F, n_max, 64)).cuda()
    d_xy = torch.from_numpy(xy.reshape(2 * F, n_max, 2)).cuda()
    d_count = torch.from_numpy(count.reshape(2 * F)).cuda()
    f0, f1 = pipeline.consecutive_pairs(F)
    P = len(f0)
    seeds = 1000 + first_pair + np.arange(P)                  # np.random.seed(1000 + pair index)
    trk = pipeline.ShardTracker(max(F, 1), max(P, 1), n_max, hot_path_config(args), n_chunks=args.chunks, taper=args.taper,
                                serial_match=None if args.serial_match < 0 else bool(args.serial_match))

    n_compose = F_total - 1                                   # rank 0 composes the first movie's trajectory
    rel_buf = np.empty((n_compose + 1, 3, 4))
    rel_buf[0] = np.eye(3, 4)                                 # the first camera (M1)

    def step():
        # results stay where the library wrote them (page-locked host memory); the step consumes them before the next run
        res = trk.run(d_desc, d_xy, d_count, f0, f1, seeds, unpack_mask=False, copy=False)
        # the one exchange of the path: relative poses + counters to every rank (RCCL all-gather over xGMI)
        rel_all, _ = sdist.gather_shard_results(res.pose, res.stats, pairs_total)
        glob = None
        if rk.rank == 0:
            # host-side composition (utils/utills.py:749-757); pairs with no pose keep the previous camera
            rel_buf[1:] = rel_all[:n_compose]
            bad = ~np.isfinite(rel_buf.reshape(n_compose + 1, 12)).all(axis=1)
            if bad.any():
                rel_buf[bad] = np.eye(3, 4)
            glob = pipeline.compose_global(rel_buf)
        return res, glob

    for _ in range(args.warmup):
        step()
    rk.fence()
    match_ms, stage_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, glob = step()
        match_ms.append(res.timing_ms[0])
        stage_ms.append(res.timing_ms.copy())
    rk.fence()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0)
    # the last step's results, detached from the tracker's output buffers (later legs run the tracker again)
    res = dataclasses.replace(res, pose=res.pose.copy(), stats=res.stats.copy(), mask=res.mask.copy())
    return dict(elapsed=elapsed, pairs_total=pairs_total, match_ms=match_ms, stage_ms=stage_ms, res=res, glob=glob, seq=seq,
                desc=desc, xy=xy, count=count, trk=trk, f0=f0, f1=f1, seeds=seeds, gen_s=gen_s, F=F, P=P, n_max=n_max,
                scaling=scaling)


def matcher_roofline(args, leg):
    """Roofline of the dominant kernel (all-pairs matcher) from HIP events on the launch stream inside slam_track_pairs."""
    cnt = leg["count"].astype(np.int64)
    pairs_stereo = int((cnt[:, 0] * cnt[:, 1]).sum())
    pairs_temporal = int((cnt[:-1, 0] * cnt[1:, 0]).sum())
    desc_pairs = pairs_stereo + pairs_temporal                    # descriptor pairs per step on this rank
    m_ms = float(np.mean(leg["match_ms"]))
    exec_rate = desc_pairs * EXEC_OPS_PER_PAIR / (m_ms * 1e-3) / 1e12
    alg_bytes = int(cnt.sum()) * 64 * 1 + (2 * leg["F"] - 1) * leg["n_max"] * 4   # every descriptor row once + the key arrays
    slots = desc_pairs * ISSUE_SLOTS_PER_PAIR / (m_ms * 1e-3) / 1e12
    roofline = {
        "kernel": "allpairs_top2_kernel (stereo + temporal problems, one launch per chunk)",
        "bound": "valu", "unit": "Tlaneop/s",
        "achieved": round(exec_rate, 3), "peak": round(VALU_PEAK_TLANEOPS, 2),
        "frac": round(exec_rate / VALU_PEAK_TLANEOPS, 4),
        "ops_per_descriptor_pair_executed": EXEC_OPS_PER_PAIR,
        "issue_slots_per_descriptor_pair": ISSUE_SLOTS_PER_PAIR,
        "issue_slot_rate_Tslots": round(slots, 3),
        "issue_slot_frac_of_nominal_peak": round(slots / VALU_PEAK_TLANEOPS, 4),
        "issue_slot_frac_of_measured_sustained": round(slots / VALU_SUSTAINED_TSLOTS, 4),
        "descriptor_pairs_Tps": round(desc_pairs / (m_ms * 1e-3) / 1e12, 4),
        "measured_instruction_mix_ceiling_Tpairs": round(MIX_CEILING_TPAIRS, 4),
        "frac_of_measured_instruction_mix_ceiling": round(desc_pairs / (m_ms * 1e-3) / 1e12 / MIX_CEILING_TPAIRS, 4),
        "achieved_survey_units": round(desc_pairs * SURVEY_OPS_PER_PAIR / (m_ms * 1e-3) / 1e12, 3),
        "descriptor_pairs_per_step": desc_pairs, "kernel_ms_per_step": round(m_ms, 4),
        "hbm_algorithmic_bytes_per_step": alg_bytes,
        "hbm_algorithmic_GBps": round(alg_bytes / (m_ms * 1e-3) / 1e9, 2), "hbm_peak_GBps": HBM_PEAK_GBPS,
        "traffic": None,
    }
    roofline.update(pmc_traffic(args, m_ms))
    return roofline


def stage_dict(stage_ms):
    stages = np.mean(np.array(stage_ms), axis=0)
    names = ("match", "join", "triangulate", "ransac_rounds", "mask_refit", "call_wall", "host_rng_replay")
    return {n: round(float(stages[i]), 3) for i, n in enumerate(names)}


def run_trajectory(args, rk: Rank):
    world = rk.world
    scaling = args.scaling
    if scaling == "auto":
        scaling = "strong"                      # at N = 1 the two legs are the same workload; the series is strong
    leg = trajectory_leg(args, rk, scaling)
    weak = None
    if args.scaling == "auto" and world > 1:
        # free the strong leg's device tensors before the weak leg allocates a whole movie per rank
        keep = {k: leg[k] for k in ("elapsed", "pairs_total", "match_ms", "stage_ms", "res", "glob", "seq", "count", "gen_s",
                                     "F", "P", "n_max", "scaling")}
        leg = keep
        rk.torch.cuda.empty_cache()
        wl = trajectory_leg(args, rk, "weak")
        weak = {"value": round(wl["pairs_total"] * args.steps / wl["elapsed"], 2), "unit": "frame-pairs/s",
                "ms_per_step": round(1e3 * wl["elapsed"] / args.steps, 3), "frame_pairs_per_step": wl["pairs_total"],
                "workload": f"every rank its own {args.frames}-frame range of a longer drive ({world} x {args.frames - 1} pairs per step)"}
        if rk.rank == 0:
            weak["matcher_kernel_ms_per_step_rank0"] = round(float(np.mean(wl["match_ms"])), 4)
        del wl
    if rk.rank != 0:
        return
    F_total = args.frames
    res, glob, seq = leg["res"], leg["glob"], leg["seq"]
    ms_per_step = 1e3 * leg["elapsed"] / args.steps
    value = leg["pairs_total"] * args.steps / leg["elapsed"]
    # trajectory quality against the generator's ground truth (slam_amd/trajectory.py: the reference's
    # between()/camera-centre series condensed into ATE / RPE), outside the timed region
    from slam_amd import trajectory
    traj = trajectory.summary(glob, np.array([seq.gt_global(i) for i in range(F_total)]))
    if scaling == "weak":
        workload = (f"synthetic KITTI-shaped stereo, {F_total} frames x {args.features} features/image per GPU "
                    f"(BASELINE configs[1]); {F_total - 1} frame pairs per GPU per step")
    elif world == 1:
        workload = (f"synthetic KITTI-shaped stereo, {F_total} frames x {args.features} features/image, 1 MI355X "
                    f"(BASELINE configs[1]); {F_total - 1} frame pairs per step")
    else:
        workload = (f"synthetic KITTI-shaped stereo, ONE {F_total}-frame sequence x {args.features} features/image "
                    f"block-partitioned by frame range over {world} GPUs (BASELINE configs[2]); {leg['pairs_total']} frame pairs per step")
    out = {
        "metric": "frame-pairs/sec (match+triangulate+PnP-RANSAC)", "value": round(value, 2), "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u32 popcount + f64",
        "data": "synthetic",
        "config": {"workload": workload, "descriptor_bytes": 61, "norm": "NORM_HAMMING2",
                   "match": {"ocv_legacy": "cross-check brute force", "mutual": "cross-check brute force (mutual rule)",
                             "knn": "kNN (k = 2) + ratio 0.6"}[args.match_rule],
                   "ransac": "adaptive (prob 0.9999, cap 1000), seeds 1000+pair", "parallelism": f"frame-range x{world}"},
        "roofline": matcher_roofline(args, leg),
        "stage_ms": stage_dict(leg["stage_ms"]),
        "quality": {"ate_rmse_m_vs_ground_truth": round(traj["ate_rmse_m"], 4),
                    "end_drift_pct_of_path": round(traj["end_drift_pct_of_path"], 4),
                    "rpe_trans_pct": round(traj["rpe_trans_pct"], 4),
                    "rpe_rot_deg_per_m": round(traj["rpe_rot_deg_per_m"], 6), "pairs_ok": int(res.ok.sum()), "pairs": leg["P"],
                    "mean_tracked": round(float(res.tracked.mean()), 1),
                    "mean_supporters": round(float(res.supporters.mean()), 1),
                    "mean_iterations": round(float(res.stats[:, 2].mean()), 2)},
        "setup_s": {"synthetic_generation": round(leg["gen_s"], 2)},
    }
    if world > 1:
        out["roofline"]["note"] = "rank 0's launches (its block of the sequence)"
    if weak is not None:
        out["weak"] = weak
    if world == 1 and not args.no_ingest and "desc" in leg:
        out["pcie_inclusive"] = pcie_inclusive(args, leg)
    if world == 1 and not args.no_dropin and "desc" in leg:
        out["dropin_whole_movie"] = dropin_whole_movie(args, leg)
    if world == 1 and not args.no_cpu_baseline and "desc" in leg:
        out["cpu_baseline"] = cpu_baseline(args, leg)
    rk.emit(out)


def pcie_inclusive(args, leg):
    """The same step with the shard's detections starting in pinned HOST memory (slam_amd/ingest.py): the H2D copy
    alone, the step with the copy in front of it, and the step with the next shard's copy overlapped (double
    buffered) -- in the padded hand-over format and in the dense wire format (61-byte rows, no empty rows; unpacked on
    the device by slam_unpack_detections).  Reported beside `value`, never as it: `value` is quoted with detections
    resident in HBM."""
    import torch
    from slam_amd import ingest
    desc, xy, count, trk, f0, f1, seeds = (leg[k] for k in ("desc", "xy", "count", "trk", "f0", "f1", "seeds"))
    F, n_max = len(count), desc.shape[2]
    P = len(f0)
    n = max(2, min(args.steps, 4))

    def measure(st):
        def run(k):
            d = st.device(k)
            trk.run(*d, f0, f1, seeds, unpack_mask=False)
            st.release(k)

        st.upload(0); st.upload(1); torch.cuda.synchronize()                  # first touch of the device slots
        t0 = time.perf_counter()
        for k in range(n):
            st.upload(k)
        torch.cuda.synchronize()
        h2d_ms = 1e3 * (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for k in range(n):                                                    # copy, then compute, nothing overlapped
            st.upload(k)
            torch.cuda.synchronize()
            run(k)
        torch.cuda.synchronize()
        serial_ms = 1e3 * (time.perf_counter() - t0) / n
        st.upload(0)
        torch.cuda.synchronize()                                              # pipeline filled: steady state of a stream of shards
        t0 = time.perf_counter()
        for k in range(n):                                                    # shard k+1 uploads while shard k is tracked
            st.upload(k + 1)
            run(k)
        torch.cuda.synchronize()
        overlap_ms = 1e3 * (time.perf_counter() - t0) / n
        gb = st.bytes_per_shard() / 1e9
        return {"h2d_ms_per_shard": round(h2d_ms, 3), "h2d_GBps": round(gb / (h2d_ms * 1e-3), 2), "shard_GB": round(gb, 3),
                "serial_ms_per_step": round(serial_ms, 3), "serial_frame_pairs_per_s": round(P / (serial_ms * 1e-3), 1),
                "overlapped_ms_per_step": round(overlap_ms, 3),
                "overlapped_frame_pairs_per_s": round(P / (overlap_ms * 1e-3), 1), "steps": n}

    st = ingest.ShardStager(F, n_max)
    for k in range(2):
        s = st.host_slot(k)
        s.desc[:], s.xy[:], s.count[:] = desc, xy, count
    out = measure(st)
    del st
    torch.cuda.empty_cache()
    dst = ingest.DenseShardStager(F, n_max, width=61)
    for k in range(2):
        s = dst.host_slot(k)
        s.reset()
        for j in range(F):
            for side in range(2):
                c = int(count[j, side])
                s.put(j, side, xy[j, side, :c], desc[j, side, :c, :61])
    out["dense_wire_format"] = measure(dst)
    return out


def dropin_whole_movie(args, leg):
    """The reference-shaped call: find_features_in_consecutive_frames_whole_movie() + DataBase(...) exactly as
    DataDirectory/Data.py:42-43 issues them, on the same movie, detections starting in host memory (as the reference's
    own detections do).  Outside the timed region of `value`; one warm call, then one timed call."""
    from slam_amd import utills
    from slam_amd.types import DataBase
    utills.set_detection_source(utills.ArrayDetections(leg["desc"], leg["xy"], leg["count"]))
    out = {}
    for label in ("first_call_s", "s"):
        np.random.seed(7)
        t0 = time.perf_counter()
        feats, percs, glob, rel = utills.find_features_in_consecutive_frames_whole_movie()
        t1 = time.perf_counter()
        db = DataBase(feats, percs, glob, rel)
        t2 = time.perf_counter()
        out[label] = round(t2 - t0, 3)
    out.update({"find_features_whole_movie_s": round(t1 - t0, 3), "database_ctor_s": round(t2 - t1, 3),
                "frames": len(leg["count"]), "tracks": int(db.get_num_tracks()), "database": type(db).__name__,
                "call": "utills.find_features_in_consecutive_frames_whole_movie() + DataBase(...) (Data.py:42-43)"})
    utills.set_detection_source(None)
    return out


def pmc_traffic(args, live_kernel_ms):
    """HBM bytes per matcher launch from the committed PMC passes (tools/profile_gpu.sh: separate FETCH_SIZE and
    WRITE_SIZE runs of this same workload, one launch per step; FETCH_SIZE doubled as the gfx950 guide prescribes).
    PMC counters cannot be collected from inside a timed run, so the figure is the profile's, valid only for the
    default workload; any other size reports null.  `traffic_profile_kernel_ms` is the kernel's duration in the trace
    that belongs to the PMC passes: if the live figure has moved away from it the kernel has changed since the profile
    was taken and `traffic_stale` says so."""
    path = os.path.join(ROOT, PMC_SUMMARY)
    if not os.path.exists(path):
        path = os.path.join(ROOT, PMC_SUMMARY_FALLBACK)
    if (args.frames, args.features, args.seed, args.config, args.gpus) != (3450, 2000, 1234, "trajectory", 1) or not os.path.exists(path):
        return {}
    with open(path) as fh:
        pmc = json.load(fh)
    for name, e in pmc.items():
        if name.startswith("allpairs_top2_kernel") and "hbm_read_bytes_per_dispatch_corrected" in e:
            rd, wr = e["hbm_read_bytes_per_dispatch_corrected"], e.get("hbm_write_bytes_per_dispatch", 0)
            out = {"traffic": rd + wr, "traffic_unit": "bytes per whole-shard launch (PMC read x2 + write)",
                   "traffic_source": os.path.relpath(path, ROOT)}
            ref_ms = e.get("trace_avg_ms")
            if ref_ms:
                out["traffic_profile_kernel_ms"] = ref_ms
                out["traffic_stale"] = bool(abs(live_kernel_ms - ref_ms) > 0.1 * ref_ms)
            return out
    return {}


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import slam_oracle as orc
    orc.bf_match = lambda a, b, norm=7: c_oracle.match_crosscheck_w64(c_oracle.pad64(a), c_oracle.pad64(b), norm)
    if hasattr(orc, "_port_pnp"):
        orc.pnp = orc._port_pnp                    # undo _bind_cv2
    return orc


def _try_cv2():
    try:
        import cv2
        return cv2
    except Exception:                              # not installed in the build image or on the GPU box
        return None


def _bind_cv2(orc, cv2):
    """The oracle's per-pair driver with the arithmetic the reference leaves to OpenCV bound to the installed cv2, called
    as the reference calls it: cv2.BFMatcher(norm, crossCheck=True).match (utils/utills.py:65-66) and cv2.solvePnP +
    cv2.Rodrigues (:585-596, :574-582).  Everything else (rect test, join, triangulation, consensus, the adaptive loop)
    is the reference's own numpy arithmetic, restated in the oracle."""
    if not hasattr(orc, "_port_pnp"):
        orc._port_pnp = orc.pnp

    def bf_match(a, b, norm=7):
        ms = cv2.BFMatcher(int(norm), crossCheck=True).match(np.ascontiguousarray(a, dtype=np.uint8), np.ascontiguousarray(b, dtype=np.uint8))
        ms = sorted(ms, key=lambda m: m.queryIdx)
        return (np.array([m.queryIdx for m in ms], np.int32), np.array([m.trainIdx for m in ms], np.int32),
                np.array([int(round(m.distance)) for m in ms], np.int32))

    def pnp(obj, img, calib, flag):
        ok, rvec, tvec = cv2.solvePnP(np.ascontiguousarray(obj, dtype=np.float64), np.ascontiguousarray(img, dtype=np.float64),
                                      np.asarray(calib, dtype=np.float64), None, flags=int(flag))
        if not ok:
            return None
        return np.hstack([np.asarray(cv2.Rodrigues(rvec)[0], dtype=np.float64), np.asarray(tvec, dtype=np.float64).reshape(3, 1)])

    orc.bf_match, orc.pnp = bf_match, pnp
    return orc


def _oracle_pair(job, cv2=None):
    """One frame pair through the CPU oracle (module-level so that a process pool can run it); with cv2, through the
    oracle's driver bound to the installed OpenCV."""
    a, b, seed = job
    orc = _oracle()
    if cv2 is not None:
        a, b = dict(a), dict(b)
        a.pop("stereo", None)
        orc = _bind_cv2(orc, cv2)
    a["stereo"] = orc.stereo_match(a["dsc_l"], a["dsc_r"], a["xy_l"], a["xy_r"])   # carried over from the previous pair
    np.random.seed(int(seed))
    t0 = time.perf_counter()
    st = {}
    with np.errstate(all="ignore"):
        want = orc.track_pair(a, b, stats=st)
    dt = time.perf_counter() - t0
    return dt, st.get("iterations", -1), want["kp"], want["mask"], want["T"]


def cpu_baseline(args, leg):
    """The CPU oracle ("port": numpy restatement + C matcher) on a bounded sample of the same pairs, timed on this
    box's host cores: one thread (the reference is single-threaded Python) and all cores the box gives this job
    (one process per core, pairs are independent); also cross-checks the GPU results of the one-thread sample."""
    import multiprocessing as mp
    desc, xy, count, seeds, res, trk = (leg[k] for k in ("desc", "xy", "count", "seeds", "res", "trk"))
    F = len(count)
    kp = res.tracked_kp.cpu().numpy()
    mask = __import__("slam_amd.ops", fromlist=["ops"]).mask_words_to_bool(res.mask, trk.n_max)

    def frame(i):
        n0, n1 = count[i]
        return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])

    def jobs(n):
        picks = np.linspace(0, F - 2, min(n, F - 1)).astype(int)
        return picks, [(frame(p), frame(p + 1), seeds[p]) for p in picks]

    picks, js = jobs(args.cpu_pairs)
    n = len(picks)
    t_total, exact_kp, exact_mask, exact_it, worst_pose = 0.0, 0, 0, 0, 0.0
    for p, job in zip(picks, js):
        dt, it, w_kp, w_mask, w_T = _oracle_pair(job)
        t_total += dt
        exact_it += int(res.stats[p, 2] == it)
        T = len(w_kp)
        exact_kp += int(res.stats[p, 0] == T and np.array_equal(kp[p, :T], w_kp))
        exact_mask += int(np.array_equal(mask[p, :T], w_mask))
        if w_T is not None:
            worst_pose = max(worst_pose, float(np.linalg.norm(res.pose[p] - w_T) / np.linalg.norm(w_T)))
    out = {"value": round(n / t_total, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
           "sample": f"{n} of {F - 1} frame pairs of the same sequence (evenly spaced), CPU oracle: C 64-bit-popcount "
                     f"cross-check matcher (one new stereo match + one temporal match per pair, as the reference's "
                     f"steady state) + numpy triangulation / AP3P / consensus / LM refit",
           "parity_on_sample": {"tracked_sets_bit_exact": f"{exact_kp}/{n}", "supporter_masks_bit_exact": f"{exact_mask}/{n}",
                                "adaptive_iteration_counts_equal": f"{exact_it}/{n}",
                                "worst_pose_rel_err": float(f"{worst_pose:.3e}"),
                                "note": "a pair can differ only where a draw repeats its fourth index and AP3P picks its "
                                        "root on rounding noise (DESIGN.md section 2; over the whole movie 13 of 3449 masks, 74 of 3449 iteration counts)"},
           "cpu_model": _cpu_model(), "host_cpus_visible": os.cpu_count()}
    cv2 = _try_cv2()
    if cv2 is not None:
        out["cv2_reference_path"] = _cv2_leg(cv2, picks, js, res, kp, mask)
    _oracle()                                                              # the port's own bindings again
    # all cores: the job's CPU share (OMP_NUM_THREADS is set to it on the GPU box; os.cpu_count() shows the whole machine)
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or len(os.sched_getaffinity(0)), 16))
    if cores > 1 and args.cpu_pairs_all_cores > 0:
        _, js = jobs(args.cpu_pairs_all_cores)
        os.environ["OMP_NUM_THREADS"] = "1"                                # one BLAS thread per worker process
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(_warm, range(cores))                                  # import numpy / load the C matcher
            t0 = time.perf_counter()
            pool.map(_oracle_pair, js, chunksize=1)
            wall = time.perf_counter() - t0
        os.environ["OMP_NUM_THREADS"] = str(cores)
        out["all_cores"] = {"value": round(len(js) / wall, 3), "unit": "frame-pairs/s", "cores": cores,
                            "sample": f"{len(js)} evenly spaced frame pairs over {cores} worker processes, wall clock"}
    return out


def _cv2_leg(cv2, picks, js, res, kp, mask, limit=48):
    """Only where `import cv2` succeeds ("if cv2 is importable, the reference-shaped cv2 path"): the same
    sample pairs with BFMatcher / solvePnP / Rodrigues from the installed OpenCV, one thread, and how the GPU results
    compare with THAT -- the comparison DESIGN.md section 2 calls unpinned where cv2 is absent."""
    step = max(1, len(picks) // limit)
    t_total, n, eq_kp, eq_mask, worst = 0.0, 0, 0, 0, 0.0
    try:
        threads = cv2.getNumThreads()
        cv2.setNumThreads(1)
    except Exception:
        threads = None
    for p, job in list(zip(picks, js))[::step]:
        dt, _, w_kp, w_mask, w_T = _oracle_pair(job, cv2=cv2)
        t_total += dt
        n += 1
        T = len(w_kp)
        same = bool(res.stats[p, 0] == T and np.array_equal(kp[p, :T], w_kp))
        eq_kp += same
        eq_mask += bool(same and np.array_equal(mask[p, :T], w_mask))
        if same and w_T is not None and np.isfinite(res.pose[p]).all():
            worst = max(worst, float(np.linalg.norm(res.pose[p] - w_T) / np.linalg.norm(w_T)))
    return {"value": round(n / t_total, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "reference",
            "cv2_version": getattr(cv2, "__version__", "?"), "cv2_threads_before": threads,
            "sample": f"{n} of the one-thread sample's pairs; cv2.BFMatcher(NORM_HAMMING2, crossCheck=True) + cv2.solvePnP "
                      f"(AP3P, ITERATIVE) + cv2.Rodrigues inside the reference's loop structure",
            "gpu_vs_cv2": {"tracked_sets_equal": f"{eq_kp}/{n}", "supporter_masks_equal": f"{eq_mask}/{n}",
                           "worst_pose_rel_err": float(f"{worst:.3e}")}}


def _warm(_):
    _oracle()
    return 0


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------------ other configs
def _timed_passes(rk, trk, dev_args, f0, f1, seeds, args):
    for _ in range(args.warmup):
        trk.run(*dev_args, f0, f1, seeds, unpack_mask=False)
    rk.fence()
    t0, tm = time.perf_counter(), []
    for _ in range(args.steps):
        res = trk.run(*dev_args, f0, f1, seeds, unpack_mask=False)
        tm.append(res.timing_ms.copy())
    rk.fence()
    return res, rk.max_over_ranks(time.perf_counter() - t0), tm


def run_stress(args, rk: Rank):
    """BASELINE configs[3]: 8k features per image, 60 % outliers among the tracked correspondences, 10 000 RANSAC
    hypotheses per frame pair (adaptive stop disabled so the work is defined)."""
    torch = rk.torch
    from slam_amd import pipeline
    from slam_amd.synth import SynthSequence, stress_config
    F = args.frames
    scfg = stress_config(n_frames=F + 20, n_features=args.features)
    desc, xy, count = SynthSequence(scfg).frames(10, 10 + F)
    n_max = scfg.n_max
    hp = hot_path_config(args, fixed_iterations=10000)
    f0, f1 = pipeline.consecutive_pairs(F)
    P = F - 1
    trk = pipeline.ShardTracker(F, P, n_max, hp)
    dev_args = (torch.from_numpy(desc.reshape(2 * F, n_max, 64)).cuda(), torch.from_numpy(xy.reshape(2 * F, n_max, 2)).cuda(),
                torch.from_numpy(count.reshape(2 * F)).cuda())
    seeds = 7000 + np.arange(P)
    res, elapsed, tm = _timed_passes(rk, trk, dev_args, f0, f1, seeds, args)
    if rk.rank != 0:
        return
    T = res.stats[:, 0].astype(np.int64)
    tests = float((res.stats[:, 3].astype(np.int64) * T).sum())            # point-hypothesis tests per pass
    st = stage_dict(tm)
    score_ms = st["ransac_rounds"]
    flops = 50.0 * tests                                                    # ~50 FP64 flop per point and hypothesis
    out = {"metric": "frame-pairs/sec (match+triangulate+PnP-RANSAC)", "value": round(P * args.steps / elapsed, 2),
           "unit": "frame-pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32 popcount + f64", "data": "synthetic",
           "config": {"workload": f"stress (BASELINE configs[3]): {F} frames x {args.features} features/image, "
                                  f"{100 * (1 - float(res.stats[:, 1].sum()) / max(1, int(T.sum()))):.0f} % outliers among tracked "
                                  f"correspondences, 10 000 counted RANSAC hypotheses per pair; {P} frame pairs per step",
                      "norm": "NORM_HAMMING2", "match": "cross-check brute force", "ransac": "fixed 10 000 iterations"},
           "roofline": {"kernel": "score_lanes_f32_kernel / score_f32_kernel (consensus scoring, RANSAC rounds)", "bound": "fp32 valu",
                        "unit": "TFLOP/s", "achieved": round(flops / (score_ms * 1e-3) / 1e12, 3), "peak": 157.3,
                        "frac": round(flops / (score_ms * 1e-3) / 1e12 / 157.3, 4),
                        "note": "50 flop per point-hypothesis test over the RANSAC-round stage time (sampling + AP3P + scoring + D2H); "
                                "tests are pre-classified in single precision behind an error bound and settled in FP64 where "
                                "undecided (counts identical to the FP64 kernels, whose peak was 78.6 and frac 0.35)",
                        "point_hypothesis_tests_per_step": tests, "traffic": None},
           "stage_ms": st,
           "quality": {"mean_tracked": round(float(T.mean()), 1), "min_tracked": int(T.min()),
                       "mean_supporters": round(float(res.stats[:, 1].mean()), 1),
                       "iterations": int(res.stats[0, 2]), "mean_draws": round(float(res.stats[:, 3].mean()), 1),
                       "pairs_ok": int(res.ok.sum()), "pairs": P,
                       "point_hypothesis_tests_per_s": float(f"{tests * args.steps / elapsed:.4g}")}}
    rk.emit(out)


def run_loops(args, rk: Rank):
    """BASELINE configs[4]: loop-closure candidate validation, 3450 frames x 5 candidates each.  N = 1: one device pass.
    N > 1 (north_star: "Loop-closure candidate consensus matching ... shards the same way"): the ONE flat list of
    (candidate, current) pairs block-partitioned over the ranks (slam_amd.dist.shard_list), strong scaling: `value` = all
    pairs / max-over-ranks time.  Two plans (slam_amd.dist.track_pairs_sharded): "one" -- a rank holds, stereo-matches and
    tracks the frames its block touches, no data-path collective; "two" -- every frame is stereo-matched on ONE rank
    (block partition of the frames), the keys are all-gathered (28 MB: the path's one exchange step, RCCL over xGMI) and
    the pair blocks run on them, temporal problems only.  Either way one all-gather of the 128-byte pose + counter rows
    ends the step."""
    torch = rk.torch
    from slam_amd import dist as sdist
    from slam_amd import pipeline
    from slam_amd.synth import SynthConfig, SynthSequence, loop_candidate_pairs
    F = args.frames
    scfg = SynthConfig(n_frames=F, n_features=args.features, seed=args.seed)
    n_max = scfg.n_max
    g0, g1 = loop_candidate_pairs(F)                                        # movie frame indices
    P_total = len(g0)
    two_auto, cost_one, cost_two = sdist.pair_list_plan(g0, g1, F, rk.world)
    # (a one-rank process group, SLAM_BENCH_FORCE_DIST=1, takes the two-phase path when asked: it rehearses the RCCL
    # all-gather of the key table on a one-GPU box)
    two = (rk.world > 1 and (args.loops_plan == "two" or (args.loops_plan == "auto" and two_auto))) or \
          (rk.use_dist and args.loops_plan == "two")
    a, b = sdist.shard_list(P_total, rk.rank, rk.world)
    used, f0, f1 = sdist.frames_of_block(g0, g1, a, b)
    fa, fb = sdist.block_range(F, rk.rank, rk.world)
    if two:
        # the rank holds its frame block (phase 1) and the frames its pair block touches (phase 2) as ONE ascending list;
        # the frame block is a contiguous run of it
        used = np.union1d(used, np.arange(fa, fb))
        f0, f1 = np.searchsorted(used, g0[a:b]).astype(np.int32), np.searchsorted(used, g1[a:b]).astype(np.int32)
    seq = SynthSequence(scfg)
    t0 = time.time()
    Fl = len(used)
    desc = np.zeros((max(Fl, 1), 2, n_max, 64), np.uint8)
    xy = np.zeros((max(Fl, 1), 2, n_max, 2), np.float32)
    count = np.zeros((max(Fl, 1), 2), np.int32)
    if Fl == F:
        desc, xy, count = seq.frames(0, F)
    else:
        for k, f in enumerate(used):                                        # every frame is a pure function of its index
            fr = seq.frame(int(f))
            desc[k], xy[k], count[k] = fr["desc"], fr["xy"], fr["count"]
    gen_s = time.time() - t0
    P = b - a
    Fd = len(count)
    hp = hot_path_config(args)
    trk = pipeline.ShardTracker(Fd, max(P, 1), n_max, hp, n_chunks=1)
    dev_args = (torch.from_numpy(desc.reshape(2 * Fd, n_max, 64)).cuda(), torch.from_numpy(xy.reshape(2 * Fd, n_max, 2)).cuda(),
                torch.from_numpy(count.reshape(2 * Fd)).cuda())
    seeds = 9000 + a + np.arange(P)                                         # a pair's seed is its index in the whole list
    i0 = int(np.searchsorted(used, fa)) if two else 0                       # the frame block's place in the rank's list
    used_dev = torch.from_numpy(np.asarray(used, np.int64)).cuda()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    phase_ms = []

    def step():
        keys = None
        if two:
            ev[0].record()
            table = sdist.stereo_keys_sharded(
                F, lambda x, y: pipeline.stereo_keys(dev_args[0][2 * i0:2 * (i0 + y - x)], dev_args[2][2 * i0:2 * (i0 + y - x)], hp))
            ev[1].record()
            keys = table.to(used_dev.device)[used_dev].contiguous()
            ev[2].record()
        res = trk.run(*dev_args, f0, f1, seeds, unpack_mask=False, stereo_key=keys, copy=False)   # views: consumed before the next run
        if two:
            phase_ms.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
        pose, stats = sdist.gather_shard_results(res.pose, res.stats, P_total)
        return res, stats

    for _ in range(args.warmup):
        step()
    rk.fence()
    t0, tm = time.perf_counter(), []
    for _ in range(args.steps):
        res, stats = step()
        tm.append(res.timing_ms.copy())
    rk.fence()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0)
    frames_per_rank = rk.max_over_ranks(float(Fl))
    if rk.rank != 0:
        return
    T = np.maximum(stats[:, 0], 1)
    perc = 100.0 * stats[:, 1] / T
    near = np.arange(P_total) % 5 < 2
    cnt = count.astype(np.int64)
    n_stereo = (fb - fa) if two else Fl                                     # stereo problems of rank 0 per step
    s_lo = i0 if two else 0
    desc_pairs = int((cnt[s_lo:s_lo + n_stereo, 0] * cnt[s_lo:s_lo + n_stereo, 1]).sum() + (cnt[f0, 0] * cnt[f1, 0]).sum())
    st = stage_dict(tm)
    if two:                                                                 # phase 1 is a matcher launch of its own
        ph = np.array(phase_ms[-args.steps:])
        st["stereo_phase_match_and_allgather"] = round(float(ph[:, 0].mean()), 3)
        st["stereo_phase_gather_rows"] = round(float(ph[:, 1].mean()), 3)
    exec_rate = desc_pairs * EXEC_OPS_PER_PAIR / ((st["match"] + (st["stereo_phase_match_and_allgather"] if two else 0)) * 1e-3) / 1e12
    world = rk.world
    where = "one device pass" if world == 1 else (
        f"the ONE pair list block-partitioned over {world} GPUs, " +
        (f"two phases: every frame stereo-matched on one rank ({fb - fa} per rank), keys all-gathered, then the pair blocks "
         f"(temporal problems only; a rank holds at most {int(frames_per_rank)} of {F} frames)" if two else
         f"one phase: each rank holding and stereo-matching the frames its block touches (at most {int(frames_per_rank)} of {F})"))
    out = {"metric": "frame-pairs/sec (match+triangulate+PnP-RANSAC)", "value": round(P_total * args.steps / elapsed, 2),
           "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
           "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
           "dtype": "u32 popcount + f64", "data": "synthetic",
           "config": {"workload": f"loop-closure candidate validation (BASELINE configs[4]): {F} frames x {args.features} features/image, "
                                  f"every frame against 5 candidates (2 revisits + 3 earlier frames); {P_total} candidate pairs per step, {where}",
                      "norm": "NORM_HAMMING2", "match": "cross-check brute force", "ransac": "adaptive (prob 0.9999, cap 1000)",
                      "acceptance": "inliers % > 85 (PoseGraph.py:24)", "parallelism": f"pair-list x{world}"},
           "roofline": {"kernel": "allpairs_top2_kernel", "bound": "valu", "unit": "Tlaneop/s", "achieved": round(exec_rate, 3),
                        "peak": round(VALU_PEAK_TLANEOPS, 2), "frac": round(exec_rate / VALU_PEAK_TLANEOPS, 4),
                        "descriptor_pairs_per_step": desc_pairs, "kernel_ms_per_step": st["match"], "traffic": None},
           "stage_ms": st,
           "quality": {"revisits_over_85pct": round(float((perc[near] > 85).mean()), 4),
                       "others_over_85pct": round(float((perc[~near] > 85).mean()), 4),
                       "mean_iterations": round(float(stats[:, 2].mean()), 1), "pairs": P_total},
           "setup_s": {"synthetic_generation": round(gen_s, 2)}}
    if world > 1 or two:
        out["roofline"]["note"] = "rank 0's launches (its block of the pair list" + ("; phase 1's time includes the all-gather)" if two else ")")
        out["config"]["plan"] = {"chosen": "two" if two else "one", "asked": args.loops_plan,
                                 "matching_problems_of_the_busiest_rank": {"one_phase": int(cost_one), "two_phase": int(cost_two)}}
    rk.emit(out)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.config == "stress" and args.gpus != 1:
        raise SystemExit("--config stress is a one-GPU configuration")
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args, argv)                                     # before anything touches the GPU
    if not args.frames:
        args.frames = 31 if args.config == "stress" else 3450
    if not args.features:
        args.features = 8000 if args.config == "stress" else 2000
    rk = Rank(args)
    {"trajectory": run_trajectory, "stress": run_stress, "loops": run_loops}[args.config](args, rk)
    rk.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())

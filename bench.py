#!/usr/bin/env python
"""Benchmark of the stereo-trajectory hot path (BASELINE.json metric: frame-pairs/s for
match + triangulate + PnP-RANSAC on synthetic KITTI-shaped stereo, 3450 frames x ~2k features per image).

    python bench.py --gpus N --steps K --warmup W         (N > 1: launched by torch.distributed.run)

One "step" = one pass of the whole hot path over one shard: 3450 stereo frames (3449 frame pairs) whose
detections are already resident in HBM, through to relative poses + supporter masks on the host.  With N ranks
every rank owns its own 3450-frame range of a longer drive (weak scaling, no data-path collective); the only
exchange is the RCCL all-gather of the relative poses and per-pair counters before rank 0 composes the global
trajectory, as north_star describes.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
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
VALU_SUSTAINED_TSLOTS = 62.0      # measured: v_xor / v_bitop3 / v_add sustained, profiles/r1_valu_peak_microbench.txt
SURVEY_OPS_PER_PAIR = 80


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=3450)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--cpu-pairs", type=int, default=24, help="frame pairs timed through the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--chunks", type=int, default=2, help="frame-range chunks pipelined over two HIP streams per GPU")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): every rank owns its own --frames range of a longer drive; strong: the one "
                         "--frames sequence is block-partitioned over the ranks (BASELINE configs[2] read literally)")
    ap.add_argument("--first-chunk", type=int, default=0, help="hypotheses evaluated speculatively per pair in RANSAC round 0 (0 = library default)")
    ap.add_argument("--taper", type=float, default=1.0, help="each chunk's size relative to the previous one (1 = equal)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the PCIe-inclusive leg (reported beside, never as, value)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from slam_amd import pipeline
    from slam_amd.config import HotPathConfig
    from slam_amd.synth import SynthConfig, SynthSequence

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SLAM_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives run on CPU tensors); the real runs use RCCL, one rank per GPU.
    backend = os.environ.get("SLAM_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count()) if backend != "nccl" else local
    torch.cuda.set_device(local)
    # Rank 0 prints ONE JSON line: RCCL writes a version banner to stdout when a communicator is created, so file
    # descriptor 1 points at stderr for the run and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # SLAM_BENCH_FORCE_DIST=1 under a one-process torch.distributed.run: the process group and the gather run at world
    # size 1 too, which rehearses the RCCL calls of the multi-GPU run on a one-GPU box.
    use_dist = world > 1 or (os.environ.get("SLAM_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    # ---- synthetic detections of this rank's frame range (outside the timed region) ----
    from slam_amd import dist as sdist
    F_total = args.frames
    if args.scaling == "strong":
        # one movie for the whole job; rank r holds the frames of its block of pairs plus one halo frame
        pa, pb, fa, fb = sdist.shard_frames(F_total, rank, world)
        scfg = SynthConfig(n_frames=F_total, n_features=args.features, seed=args.seed)
        pairs_total, first_pair = F_total - 1, pa
    else:
        fa, fb = 0, F_total
        scfg = SynthConfig(n_frames=F_total, n_features=args.features, seed=args.seed + 7919 * rank)
        pairs_total, first_pair = world * (F_total - 1), rank * (F_total - 1)
    seq = SynthSequence(scfg)
    t0 = time.time()
    desc, xy, count = seq.frames(fa, fb)
    gen_s = time.time() - t0
    F = fb - fa
    n_max = scfg.n_max
    d_desc = torch.from_numpy(desc.reshape(2 * F, n_max, 64)).cuda()
    d_xy = torch.from_numpy(xy.reshape(2 * F, n_max, 2)).cuda()
    d_count = torch.from_numpy(count.reshape(2 * F)).cuda()
    f0, f1 = pipeline.consecutive_pairs(F)
    P = len(f0)
    seeds = 1000 + first_pair + np.arange(P)          # np.random.seed(1000 + pair index)
    cfg = HotPathConfig(first_chunk=args.first_chunk) if args.first_chunk else HotPathConfig()
    trk = pipeline.ShardTracker(F, P, n_max, cfg, n_chunks=args.chunks, taper=args.taper)

    n_compose = F_total - 1                           # rank 0 composes the first movie's trajectory
    rel_buf = np.empty((n_compose + 1, 3, 4))
    rel_buf[0] = np.eye(3, 4)                         # the first camera (M1)

    def step():
        res = trk.run(d_desc, d_xy, d_count, f0, f1, seeds, unpack_mask=False)
        # the one exchange of the path: relative poses + counters to every rank (RCCL all-gather over xGMI)
        rel_all, _ = sdist.gather_shard_results(res.pose, res.stats, pairs_total)
        glob = None
        if rank == 0:
            # host-side composition (utils/utills.py:749-757); pairs with no pose keep the previous camera
            rel_buf[1:] = rel_all[:n_compose]
            bad = ~np.isfinite(rel_buf.reshape(n_compose + 1, 12)).all(axis=1)
            if bad.any():
                rel_buf[bad] = np.eye(3, 4)
            glob = pipeline.compose_global(rel_buf)
        return res, glob

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    match_ms, stage_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, glob = step()
        match_ms.append(res.timing_ms[0])
        stage_ms.append(res.timing_ms.copy())
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = pairs_total * args.steps / elapsed
        # ---- roofline of the dominant kernel (all-pairs matcher), from HIP events on the launch stream ----
        cnt = count.astype(np.int64)
        pairs_stereo = int((cnt[:, 0] * cnt[:, 1]).sum())
        pairs_temporal = int((cnt[:-1, 0] * cnt[1:, 0]).sum())
        desc_pairs = pairs_stereo + pairs_temporal                    # descriptor pairs per step (two launches)
        m_ms = float(np.mean(match_ms))
        exec_rate = desc_pairs * EXEC_OPS_PER_PAIR / (m_ms * 1e-3) / 1e12
        alg_bytes = int(cnt.sum()) * 64 * 1 + (2 * F - 1) * n_max * 4  # every descriptor row once + the key arrays
        stages = np.mean(np.array(stage_ms), axis=0)
        roofline = {
            "kernel": "allpairs_top2_kernel (stereo + temporal problems, one launch per chunk)",
            "bound": "valu", "unit": "Tlaneop/s",
            "achieved": round(exec_rate, 3), "peak": round(VALU_PEAK_TLANEOPS, 2),
            "frac": round(exec_rate / VALU_PEAK_TLANEOPS, 4),
            "ops_per_descriptor_pair_executed": EXEC_OPS_PER_PAIR,
            "issue_slots_per_descriptor_pair": ISSUE_SLOTS_PER_PAIR,
            "issue_slot_rate_Tslots": round(desc_pairs * ISSUE_SLOTS_PER_PAIR / (m_ms * 1e-3) / 1e12, 3),
            "issue_slot_frac_of_nominal_peak": round(desc_pairs * ISSUE_SLOTS_PER_PAIR / (m_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS, 4),
            "issue_slot_frac_of_measured_sustained": round(desc_pairs * ISSUE_SLOTS_PER_PAIR / (m_ms * 1e-3) / 1e12 / VALU_SUSTAINED_TSLOTS, 4),
            "achieved_survey_units": round(desc_pairs * SURVEY_OPS_PER_PAIR / (m_ms * 1e-3) / 1e12, 3),
            "descriptor_pairs_per_step": desc_pairs, "kernel_ms_per_step": round(m_ms, 4),
            "hbm_algorithmic_GBps": round(alg_bytes / (m_ms * 1e-3) / 1e9, 2), "hbm_peak_GBps": HBM_PEAK_GBPS,
            "traffic": None,
        }
        roofline.update(pmc_traffic(args))
        # trajectory quality against the generator's ground truth (slam_amd/trajectory.py: the reference's
        # between()/camera-centre series condensed into ATE / RPE), outside the timed region
        from slam_amd import trajectory
        traj = trajectory.summary(glob, np.array([seq.gt_global(i) for i in range(F_total)]))
        out = {
            "metric": "frame-pairs/sec (match+triangulate+PnP-RANSAC)", "value": round(value, 2), "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u32 popcount + f64",
            "data": "synthetic",
            "config": {"workload": (f"synthetic KITTI-shaped stereo, {F_total} frames x {args.features} features/image per GPU "
                                    f"(BASELINE configs[1]); {F_total - 1} frame pairs per GPU per step") if args.scaling == "weak"
                       else (f"synthetic KITTI-shaped stereo, ONE {F_total}-frame sequence x {args.features} features/image "
                             f"block-partitioned over {world} GPU(s) (BASELINE configs[2]); {pairs_total} frame pairs per step"),
                       "descriptor_bytes": 61, "norm": "NORM_HAMMING2", "match": "cross-check brute force",
                       "ransac": "adaptive (prob 0.9999, cap 1000), seeds 1000+pair", "parallelism": f"frame-range x{world}"},
            "roofline": roofline,
            "stage_ms": {"match": round(float(stages[0]), 3), "join": round(float(stages[1]), 3),
                         "triangulate": round(float(stages[2]), 3), "ransac_rounds": round(float(stages[3]), 3),
                         "mask_refit": round(float(stages[4]), 3), "call_wall": round(float(stages[5]), 3),
                         "host_rng_replay": round(float(stages[6]), 3)},
            "quality": {"ate_rmse_m_vs_ground_truth": round(traj["ate_rmse_m"], 4),
                        "end_drift_pct_of_path": round(traj["end_drift_pct_of_path"], 4),
                        "rpe_trans_pct": round(traj["rpe_trans_pct"], 4),
                        "rpe_rot_deg_per_m": round(traj["rpe_rot_deg_per_m"], 6), "pairs_ok": int(res.ok.sum()), "pairs": P,
                        "mean_tracked": round(float(res.tracked.mean()), 1),
                        "mean_supporters": round(float(res.supporters.mean()), 1),
                        "mean_iterations": round(float(res.stats[:, 2].mean()), 2)},
            "setup_s": {"synthetic_generation": round(gen_s, 2)},
        }
        if world == 1 and not args.no_ingest:
            out["pcie_inclusive"] = pcie_inclusive(args, desc, xy, count, trk, f0, f1, seeds)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, seq, desc, xy, count, seeds, res, trk)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def pcie_inclusive(args, desc, xy, count, trk, f0, f1, seeds):
    """The same step with the shard's detections starting in pinned HOST memory (slam_amd/ingest.py): the H2D copy
    alone, the step with the copy in front of it, and the step with the next shard's copy overlapped (double
    buffered).  Reported beside `value`, never as it: `value` is quoted with detections resident in HBM."""
    import torch
    from slam_amd import ingest
    F, n_max = len(count), desc.shape[2]
    st = ingest.ShardStager(F, n_max)
    for k in range(2):
        s = st.host_slot(k)
        s.desc[:], s.xy[:], s.count[:] = desc, xy, count
    P = len(f0)
    n = max(2, min(args.steps, 4))

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


def pmc_traffic(args):
    """HBM bytes per matcher launch from the committed PMC passes (tools/profile_gpu.sh: separate FETCH_SIZE and
    WRITE_SIZE runs of this same workload, one launch per step; FETCH_SIZE doubled as the gfx950 guide prescribes).
    PMC counters cannot be collected from inside a timed run, so the figure is the profile's, valid only for the
    default workload; any other size reports null."""
    path = os.path.join(ROOT, "profiles", "r1_final_pmc_summary.json")
    if (args.frames, args.features, args.seed) != (3450, 2000, 1234) or not os.path.exists(path):
        return {}
    with open(path) as fh:
        pmc = json.load(fh)
    for name, e in pmc.items():
        if name.startswith("allpairs_top2_kernel") and "hbm_read_bytes_per_dispatch_corrected" in e:
            rd, wr = e["hbm_read_bytes_per_dispatch_corrected"], e.get("hbm_write_bytes_per_dispatch", 0)
            return {"traffic": rd + wr, "traffic_unit": "bytes per whole-shard launch (PMC read x2 + write)",
                    "traffic_source": "profiles/r1_final_pmc_summary.json"}
    return {}


def cpu_baseline(args, seq, desc, xy, count, seeds, res, trk):
    """The CPU oracle ("port": numpy restatement + C matcher, one thread) on a bounded sample of the same pairs,
    timed on this box's host cores; also cross-checks the GPU results of those pairs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import slam_oracle as orc
    orc.bf_match = lambda a, b, norm=7: c_oracle.match_crosscheck_w64(c_oracle.pad64(a), c_oracle.pad64(b), norm)
    F = len(count)
    n = min(args.cpu_pairs, F - 1)
    picks = np.linspace(0, F - 2, n).astype(int)
    kp = res.tracked_kp.cpu().numpy()
    mask = __import__("slam_amd.ops", fromlist=["ops"]).mask_words_to_bool(res.mask, trk.n_max)

    def frame(i):
        n0, n1 = count[i]
        return dict(dsc_l=desc[i, 0, :n0, :61], dsc_r=desc[i, 1, :n1, :61], xy_l=xy[i, 0, :n0], xy_r=xy[i, 1, :n1])

    t_total, exact_kp, exact_mask, exact_it, worst_pose = 0.0, 0, 0, 0, 0.0
    for p in picks:
        a, b = frame(p), frame(p + 1)
        a["stereo"] = orc.stereo_match(a["dsc_l"], a["dsc_r"], a["xy_l"], a["xy_r"])   # carried over from the previous pair
        np.random.seed(int(seeds[p]))
        t0 = time.perf_counter()
        st = {}
        with np.errstate(all="ignore"):
            want = orc.track_pair(a, b, stats=st)
        t_total += time.perf_counter() - t0
        exact_it += int(res.stats[p, 2] == st.get("iterations", -1))
        T = len(want["kp"])
        exact_kp += int(res.stats[p, 0] == T and np.array_equal(kp[p, :T], want["kp"]))
        exact_mask += int(np.array_equal(mask[p, :T], want["mask"]))
        if want["T"] is not None:
            worst_pose = max(worst_pose, float(np.linalg.norm(res.pose[p] - want["T"]) / np.linalg.norm(want["T"])))
    return {"value": round(n / t_total, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n} of {F - 1} frame pairs of the same sequence (evenly spaced), CPU oracle: C 64-bit-popcount "
                      f"cross-check matcher (one new stereo match + one temporal match per pair, as the reference's "
                      f"steady state) + numpy triangulation / AP3P / consensus / LM refit",
            "parity_on_sample": {"tracked_sets_bit_exact": f"{exact_kp}/{n}", "supporter_masks_bit_exact": f"{exact_mask}/{n}",
                                 "adaptive_iteration_counts_equal": f"{exact_it}/{n}",
                                 "worst_pose_rel_err": float(f"{worst_pose:.3e}")}}


if __name__ == "__main__":
    main()

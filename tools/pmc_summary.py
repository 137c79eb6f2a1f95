"""Condense rocprofv3 output directories (tools/profile_gpu.sh) into the files committed under profiles/.

    python tools/pmc_summary.py gpurun_out/r1_final profiles/r1_final

Writes <prefix>_kernel_stats_<run>.csv (copied kernel_stats of each trace run) and <prefix>_pmc_summary.json: per kernel,
the counters summed over its dispatches, the number of dispatches, and the HBM traffic per dispatch following
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so reads are doubled; WRITE_SIZE is taken as is."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def short_name(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name).strip()


def main(src, prefix):
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    for run in sorted(glob.glob(os.path.join(src, "trace_*"))):
        if not os.path.isdir(run):
            continue
        for f in glob.glob(os.path.join(run, "**", "*_kernel_stats.csv"), recursive=True):
            shutil.copy(f, f"{prefix}_kernel_stats_{os.path.basename(run)[6:]}.csv")
    per = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for run in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(run):
            continue
        for f in glob.glob(os.path.join(run, "**", "*_counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short_name(row["Kernel_Name"])
                    per[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    disp[(k, os.path.basename(run))].add(row["Dispatch_Id"])
    # average duration of every kernel in the single-lane trace that belongs to these PMC passes (bench.py compares its
    # live figure with it and flags the traffic number as stale when the kernel has changed since)
    trace_ms = {}
    for f in glob.glob(os.path.join(src, "trace_chunks1", "**", "*_kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                trace_ms[short_name(row["Name"])] = round(float(row["AverageNs"]) / 1e6, 4)
    out = {}
    for k, c in per.items():
        if "at::native" in k or k.startswith("at::") or "elementwise" in k:
            continue
        n = max(len(v) for (kk, _), v in disp.items() if kk == k)
        e = {"dispatches": n}
        if k in trace_ms:
            e["trace_avg_ms"] = trace_ms[k]
        e.update({name: val for name, val in sorted(c.items())})
        if "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0 and "SQ_ACTIVE_INST_VALU" in c:
            # share of resident-wave time spent issuing VALU instructions
            e["valu_issue_share_of_wave_cycles"] = round(c["SQ_ACTIVE_INST_VALU"] / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0), 4)
        if "FETCH_SIZE" in c:
            e["hbm_read_bytes_per_dispatch_corrected"] = round(c["FETCH_SIZE"] * 1024.0 * 2.0 / n)
        if "WRITE_SIZE" in c:
            e["hbm_write_bytes_per_dispatch"] = round(c["WRITE_SIZE"] * 1024.0 / n)
        out[k] = e
    with open(f"{prefix}_pmc_summary.json", "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v.get("dispatches") for k, v in out.items()}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

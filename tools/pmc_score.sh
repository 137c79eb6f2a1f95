#!/bin/bash
# PMC counters of the scoring micro-benchmark's kernels (GPU box, repo root):  bash tools/pmc_score.sh "stress clean"
set -e
case=${1:-stress clean}
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/pmc_score
cd /tmp && export TMPDIR=/tmp
rm -rf $out
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $out -- python3 $repo/tools/bench_score.py "$case" > $out.log 2>&1 || true
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, c in acc.items():
    if "score" not in k: continue
    d = n[k] or 1
    print(k, "dispatches", d, {a: round(b / d) for a, b in c.items()})
    if c.get("SQ_WAVE_CYCLES"):
        print("   VALU issue share of wave cycles", round(4 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 3), " wait share", round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
              " LDS instr per VALU instr", round(c["SQ_INSTS_LDS"] / c["SQ_INSTS_VALU"], 3))
PY

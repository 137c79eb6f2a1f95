#!/bin/bash
# Per-kernel time of one bench configuration: rocprofv3 kernel trace + stats, top kernels printed.
#   bash tools/kernel_stats.sh <config> [extra bench.py flags]      (run on the GPU box, from the repo root)
set -e
cfg=${1:-trajectory}; shift || true
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/kstats_$cfg
cd /tmp && export TMPDIR=/tmp
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $repo/bench.py --config $cfg --steps 10 --warmup 2 --no-dropin "$@" > /dev/null 2>&1
f=$(find $out -name '*kernel_stats.csv' | sort | sed -n 1p)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s}  total {float(r['TotalDurationNs'])/1e6:9.3f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us")
PY

#!/bin/bash
# Kernel trace of a bench run + the timeline of one step (tools/step_timeline.py).  GPU box, repo root.
#   bash tools/trace_step.sh <out dir under gpurun_out> [bench.py flags]
set -e
name=${1:-trace}; shift || true
repo=${GRAFT_REPO_ROOT:-$(pwd)}
out=$repo/gpurun_out/$name
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $repo/bench.py --steps 6 --warmup 2 --no-dropin --no-cpu-baseline --no-ingest "$@" > $out/bench.json 2> $out/err.txt
cd $repo
python tools/step_timeline.py $out 3 ${LANES:-2}

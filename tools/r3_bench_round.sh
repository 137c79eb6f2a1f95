#!/bin/bash
# Round-3 measurement set (run on the GPU box through gpurun from the repo root): the default bench line, a rank's
# share at N = 8 (433 frames), the loop-closure configuration on one rank and split over two (gloo, one card), and the
# rocprofv3 command whose exit aborted in round 2.  Every step under its own time limit, chained with &&.
set -e
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout -k 10 120 python3 bench.py --frames 433 --no-cpu-baseline --no-ingest --no-dropin > $out/bench_f433.json 2> $out/bench_f433.err
timeout -k 10 120 python3 bench.py --frames 433 --chunks 1 --no-cpu-baseline --no-ingest --no-dropin > $out/bench_f433_c1.json 2> $out/bench_f433_c1.err
timeout -k 10 200 python3 bench.py --config loops --steps 5 --warmup 2 > $out/bench_loops.json 2> $out/bench_loops.err
SLAM_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --config loops --gpus 2 --steps 5 --warmup 2 > $out/bench_loops_gloo2.json 2> $out/bench_loops_gloo2.err
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_exit -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-dropin > $out/trace_exit.json 2> $out/trace_exit.err
echo "rocprofv3 rc=$?" >> $out/trace_exit.err
echo done

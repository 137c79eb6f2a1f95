#!/bin/bash
# After a pipeline change: outputs against the previous build bit for bit (tune/lib_head.so), the GPU suite, and the
# three bench lines the change is judged on.  Run through gpurun from the repo root.
set -e
tag=${1:-r3b}
out=gpurun_out/$tag
mkdir -p $out
for mode in movie loops fixed; do
  timeout -k 10 200 python3 tools/compare_libs.py tune/lib_head.so slam_amd/lib/libslam_amd.so 300 1500 $mode >> $out/compare.txt 2>&1 || echo "(not bit-identical)" >> $out/compare.txt
done
timeout -k 10 120 python3 bench.py --frames 433 --chunks 1 --no-cpu-baseline --no-ingest --no-dropin > $out/bench_f433_c1.json 2> $out/bench_f433_c1.err
timeout -k 10 120 python3 bench.py --frames 433 --no-cpu-baseline --no-ingest --no-dropin > $out/bench_f433.json 2> $out/bench_f433.err
timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-ingest --no-dropin > $out/bench_default.json 2> $out/bench_default.err
timeout -k 10 120 python3 bench.py --chunks 1 --no-cpu-baseline --no-ingest --no-dropin > $out/bench_c1.json 2> $out/bench_c1.err
timeout -k 10 200 python3 bench.py --config loops --steps 5 --warmup 2 > $out/bench_loops.json 2> $out/bench_loops.err
timeout -k 10 700 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/gputest.log 2>&1
echo done

#!/bin/bash
# Final measurement set of round 3 (GPU box, repo root, through gpurun): traces + PMC passes (tools/profile_gpu.sh), the
# bench lines of every configuration, the scoring micro-benchmark.  Every step under its own time limit.
set -e
tag=${1:-r3f}
out=gpurun_out/$tag
bash tools/profile_gpu.sh $tag
timeout -k 10 300 python3 bench.py > $out/bench_default_full.json 2> $out/bench_default_full.err
timeout -k 10 200 python3 bench.py --frames 433 --no-cpu-baseline --no-ingest --no-dropin > $out/bench_f433.json 2> $out/bench_f433.err
timeout -k 10 300 python3 bench.py --config stress > $out/bench_stress.json 2> $out/bench_stress.err
timeout -k 10 300 python3 bench.py --config loops > $out/bench_loops.json 2> $out/bench_loops.err
timeout -k 10 300 python3 tools/bench_score.py > $out/score_bench.txt 2>&1
bash tools/kernel_stats.sh stress > $out/kstats_stress.txt 2>&1
bash tools/kernel_stats.sh loops > $out/kstats_loops.txt 2>&1
echo done

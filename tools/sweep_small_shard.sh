#!/bin/bash
# The N = 8 rank's share of configs[2] (433 frames) under different chunk plans.  GPU box, repo root.
#   bash tools/sweep_small_shard.sh > gpurun_out/small_shard.txt
set -e
run() {
  timeout -k 10 120 python bench.py --frames 433 --steps 200 --warmup 20 --no-dropin --no-cpu-baseline --no-ingest "$@" 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['stage_ms'])"
}
run --chunks 1
run --chunks 2
run --chunks 2 --taper 0.6
run --chunks 2 --taper 0.35
run --chunks 3
run --chunks 1
run --chunks 2

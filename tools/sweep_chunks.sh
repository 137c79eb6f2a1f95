#!/bin/bash
# Chunk-plan sweep of the default bench workload (run on the GPU box through gpurun, from the repo root):
#   gpurun --timeout 900 -- 'bash tools/sweep_chunks.sh r1_taper'
# One bench run per (chunks, taper); each line of gpurun_out/<tag>/summary.txt: chunks taper pairs/s ms/step match_ms
set -e
tag=${1:-taper_sweep}
out=gpurun_out/$tag
mkdir -p $out
: > $out/summary.txt
for plan in "1 1.0" "2 1.0" "2 0.7" "2 0.55" "2 0.45" "2 0.35" "2 0.25" "3 0.7" "3 0.5" "3 0.35" "4 0.5" "4 0.35"; do
  set -- $plan
  timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ingest --chunks $1 --taper $2 > $out/c$1_t$2.json 2> $out/c$1_t$2.err
  python - "$1" "$2" "$out/c$1_t$2.json" >> $out/summary.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["stage_ms"]["match"])
PY
done
cat $out/summary.txt

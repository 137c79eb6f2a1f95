#!/bin/bash
# chunk-plan sweep with the serial-matcher option (run through gpurun from the repo root)
set -e
out=gpurun_out/${1:-r3c}
mkdir -p $out
B="python3 bench.py --no-cpu-baseline --no-ingest --no-dropin"
run() { # frames chunks taper serial
  timeout -k 10 120 $B --frames $1 --chunks $2 --taper $3 --serial-match $4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('frames $1 chunks $2 taper $3 serial $4 ->', d['ms_per_step'], 'ms', d['stage_ms'])" >> $out/sweep.txt
}
for F in 433 3450; do
  run $F 1 1.0 0
  run $F 2 1.0 0
  run $F 2 1.0 1
  run $F 2 0.5 1
  run $F 2 0.25 1
  run $F 3 0.5 1
  run $F 3 0.35 1
  run $F 4 0.5 1
  run $F 4 0.7 1
  run $F 6 0.6 1
done
echo done

#!/bin/bash
# front-loaded against even B splits of the matcher on small shards (a rank's share at N = 8, 16, 32 and a quarter shard)
out=gpurun_out/r3q
mkdir -p $out
for frames in 433 217 109 864 1726; do
  for v in default even; do
    lib=""; [ $v != default ] && lib=tune/libslam_amd_$v.so
    for rep in 1 2; do
      SLAM_AMD_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-dropin --no-ingest --frames $frames --steps 40 --warmup 10 > $out/s_${frames}_$v.json 2> $out/s_${frames}_$v.err || exit 1
      python - <<PY
import json
d=json.load(open('$out/s_${frames}_$v.json')); print('$frames $v', d['ms_per_step'], d['stage_ms']['match'])
PY
    done
  done
done

#!/bin/bash
# A longer run of tests/test_gpu_fuzz.py than the suite's: more matcher cases, more pipeline seeds.  GPU box, repo root.
set -e
SLAM_FUZZ_ROUNDS=${ROUNDS:-600} timeout -k 10 500 python -u -m pytest tests/test_gpu_fuzz.py -x -q -s --timeout 480 --timeout-method thread -k "matcher" 2>&1 | tail -n 3
for seed in ${SEEDS:-101 102 103 104 105 106}; do
  SLAM_FUZZ_SEED=$seed SLAM_FUZZ_CASES=${CASES:-30} timeout -k 10 300 python -u -m pytest tests/test_gpu_fuzz.py -x -q -s --timeout 280 --timeout-method thread -k "pipeline" 2>&1 | grep -E "pipeline fuzz|passed|failed|Error" 
done

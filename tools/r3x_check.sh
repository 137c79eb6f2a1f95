# round-3 check of the single-precision scoring path: its tests, then the bench configurations that lean on scoring
set -e
mkdir -p gpurun_out/r3x
timeout -k 10 300 python -u -m pytest tests/test_gpu_score_f32.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3x/f32_test.log 2>&1
timeout -k 10 300 python bench.py --config stress --no-cpu-baseline > gpurun_out/r3x/bench_stress.json 2> gpurun_out/r3x/bench_stress.err
timeout -k 10 300 python bench.py --config loops --no-cpu-baseline > gpurun_out/r3x/bench_loops.json 2> gpurun_out/r3x/bench_loops.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-ingest --no-dropin > gpurun_out/r3x/bench_default.json 2> gpurun_out/r3x/bench_default.err
timeout -k 10 300 python bench.py --frames 433 --no-cpu-baseline --no-ingest --no-dropin > gpurun_out/r3x/bench_f433.json 2> gpurun_out/r3x/bench_f433.err

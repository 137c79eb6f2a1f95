set -e
mkdir -p gpurun_out/r3y
SLAM_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r3y/bench_gloo2.json 2> gpurun_out/r3y/bench_gloo2.err
SLAM_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --config loops --gpus 2 --steps 5 --warmup 2 > gpurun_out/r3y/bench_loops_gloo2.json 2> gpurun_out/r3y/bench_loops_gloo2.err
SLAM_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ingest --no-dropin > gpurun_out/r3y/bench_rccl1.json 2> gpurun_out/r3y/bench_rccl1.err
echo done

#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun, from the repo root):
#   gpurun --timeout 900 -- 'bash tools/profile_gpu.sh r2'
# Pass 1: kernel trace + stats of the default bench command and of the single-lane variant (--chunks 1).
# Passes 2-4: PMC counters, one block per run (SQ / TCC fetch / TCC write), never mixed with a trace domain.
# Delete a previous local gpurun_out/<tag>/ first: gpurun merges new files into it and the summaries read every file.
# Summaries are written under gpurun_out/<tag>/ ; tools/pmc_summary.py condenses them into profiles/.
set -e
tag=${1:-r2}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
# --no-ingest / --no-dropin / --no-cpu-baseline: those legs run the step again in other settings (H2D copies overlapped,
# Python drivers), which would mix other launches into the per-kernel averages; the timed region that bench.py's
# roofline is computed from has none of them.
B="python3 bench.py --no-cpu-baseline --no-ingest --no-dropin"     # default steps / warmup (20 / 5): the first launches of a process run below the sustained clock
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_default -- $B > $out/bench_default.json 2> $out/trace_default.err
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_chunks1 -- $B --chunks 1 > $out/bench_chunks1.json 2> $out/trace_chunks1.err
P="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ingest --no-dropin --chunks 1"
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $out/pmc_sq -- $P > $out/pmc_sq.json 2> $out/pmc_sq.err
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $P > $out/pmc_fetch.json 2> $out/pmc_fetch.err
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $P > $out/pmc_write.json 2> $out/pmc_write.err
echo done

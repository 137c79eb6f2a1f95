#!/bin/bash
# Matcher tile sweep.  Step 1 (build container, CPU): compile match.hip once per (rows/lane, threads, B-tile rows)
# and link each against the already built objects into tune/lib_*.so, recording the compiler's resource report
# (VGPRs, spills, occupancy) for the cross-check instantiation the bench uses:
#     bash tools/tune_matcher.sh build
# Step 2 (GPU box, through gpurun): one bench run per library, selected with SLAM_AMD_LIB:
#     gpurun --timeout 1000 -- 'bash tools/tune_matcher.sh run r1_tiles'
# tune/ is git-ignored (binaries); the summaries are copied to profiles/ by hand.
set -e
cd "$(dirname "$0")/.."
VARIANTS=${VARIANTS:-"2,256,128 3,256,128 4,128,128 4,256,64 4,256,128 4,256,256 4,512,128 5,256,128 6,256,128 8,256,128"}
EXTRA=${EXTRA:-}     # extra -D flags for every variant, e.g. EXTRA="-DSLAM_JUNROLL=2"; SUFFIX names the libraries
SUFFIX=${SUFFIX:-}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
if [ "$1" = build ]; then
  python __graft_entry__.py > /dev/null
  mkdir -p tune
  touch tune/resources.txt
  for v in $VARIANTS; do
    IFS=, read r b t <<< "$v"
    o=tune/match_r${r}_b${b}_t${t}${SUFFIX}.o
    /opt/rocm/bin/hipcc $FLAGS -DSLAM_ROWS=$r -DSLAM_BLOCK=$b -DSLAM_TILE_B=$t $EXTRA -Rpass-analysis=kernel-resource-usage \
        -c slam_amd/csrc/match.hip -o $o 2> tune/rpass_r${r}_b${b}_t${t}${SUFFIX}.txt || { echo "$v: compile failed" >> tune/resources.txt; continue; }
    objs=$(ls build/obj/*.o | grep -v match.hip.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tune/lib_r${r}_b${b}_t${t}${SUFFIX}.so $o $objs -lpthread
    python - "$v" tune/rpass_r${r}_b${b}_t${t}${SUFFIX}.txt "$SUFFIX" >> tune/resources.txt <<'PY'
import re, sys
txt = open(sys.argv[2]).read()
# the bench's kernel: allpairs_top2_kernel<true, true, false> (NORM_HAMMING2, fused cross-check, top-1)
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)
for b in blocks[1:]:
    name = b.split("\n", 1)[0]
    if "allpairs_top2_kernel" in name and "Lb1ELb0ELi" in name:
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
        print(sys.argv[1] + sys.argv[3], "VGPRs", g("VGPRs"), "AGPRs", g("AGPRs"), "ScratchSize[bytes/lane]", g(r"ScratchSize \[bytes/lane\]"),
              "VGPR spills", g("VGPRs Spill"), "occupancy[waves/SIMD]", g(r"Occupancy \[waves/SIMD\]"), "LDS", g(r"LDS Size \[bytes/block\]"))
        break
else:
    print(sys.argv[1], "kernel not found in the resource report")
PY
  done
  cat tune/resources.txt
elif [ "$1" = run ]; then
  tag=${2:-tiles}
  out=gpurun_out/$tag
  mkdir -p $out
  cp tune/resources.txt $out/resources.txt
  touch $out/summary.txt
  for v in $VARIANTS; do
    IFS=, read r b t <<< "$v"
    lib=$PWD/tune/lib_r${r}_b${b}_t${t}${SUFFIX}.so
    [ -f $lib ] || { echo "$v missing" >> $out/summary.txt; continue; }
    SLAM_AMD_LIB=$lib timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-dropin --chunks 1 \
        > $out/r${r}_b${b}_t${t}${SUFFIX}.json 2> $out/r${r}_b${b}_t${t}${SUFFIX}.err
    python - "$v" $out/r${r}_b${b}_t${t}${SUFFIX}.json "$SUFFIX" >> $out/summary.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1] + sys.argv[3], "match_ms", d["stage_ms"]["match"], "pairs/s", d["value"], "pairs_ok", d["quality"]["pairs_ok"],
      "mean_supporters", d["quality"]["mean_supporters"])
PY
  done
  cat $out/summary.txt
fi

#!/bin/bash
# slam_track_pairs with the library's HOST code under AddressSanitizer + UBSan.
#   bash tools/asan_track.sh build      here (CPU): tools/asan_track_driver from all library sources, host-instrumented
#   bash tools/asan_track.sh run        GPU box, repo root: cases from the ordinary library, then the instrumented run
set -e
repo=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -g -std=c++17 -ffp-contract=off -Wno-unused-function \
    -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined \
    -x hip $repo/tools/asan_track_driver.cpp $repo/slam_amd/csrc/*.hip $repo/slam_amd/csrc/*.cpp -o $repo/tools/asan_track_driver -lpthread
  exit 0
fi
out=/tmp/asan_cases
timeout -k 10 200 python $repo/tools/asan_track_case.py $out
ASAN_OPTIONS=detect_leaks=0 timeout -k 10 200 $repo/tools/asan_track_driver $out/case_a.bin $out/case_b.bin $out/case_c.bin $out/case_d.bin $out/case_e.bin $out/case_f.bin $out/case_g.bin

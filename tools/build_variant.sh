#!/bin/bash
# Tuning build of the library with extra -D flags:  bash tools/build_variant.sh <tag> -DNAME=VALUE ...  -> tune/libslam_amd_<tag>.so
# (use with SLAM_AMD_LIB=tune/libslam_amd_<tag>.so)
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/build/obj_$tag
mkdir -p $obj $root/tune
for s in $root/slam_amd/csrc/*.hip $root/slam_amd/csrc/*.cpp; do
  x=""; case $s in *.cpp) x="-x hip";; *score.hip) x="-fno-slp-vectorize";; esac      # per-file flags as in __graft_entry__.py
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" $x -c $s -o $obj/$(basename $s).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tune/libslam_amd_$tag.so $obj/*.o -lpthread
echo built tune/libslam_amd_$tag.so

#!/bin/bash
# lab A/B: kept register steps shared between the dimension lanes (RS = 1) against unshared ones with more LDS / workspace steps
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=14 -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-value -Wno-unused-result -DLT_NOPROBE=1"
n=0
build() {  # H K WS LS RS
  hipcc $F -DLT_H=$1 -DLT_K=$2 -DLT_WS=$3 -DLT_LS=$4 -DLT_RS=$5 -DLT_NP=1 dl_variant.hip -o bin/ab_n$((2*$1))k$2_rs$5_ws$3_ls$4 2>/dev/null &
  n=$((n+1)); if [ $((n % 7)) = 0 ]; then wait; fi
}
build 4 32 0 0 1; build 4 32 4 4 0
build 5 32 4 4 1; build 5 32 9 6 0
build 6 32 11 4 1; build 6 32 14 4 0
build 6 16 1 1 1; build 6 16 4 4 0; build 6 16 3 3 0
build 5 24 0 0 1; build 5 24 3 3 0
build 6 24 5 4 1; build 6 24 8 4 0
build 4 28 0 0 1; build 4 28 2 2 0
build 5 17 0 0 1; build 5 17 1 1 0
build 6 17 1 1 1; build 6 17 5 4 0
build 6 20 2 2 1; build 6 20 5 4 0
build 5 20 0 0 1; build 5 20 1 1 0
wait
ls bin/ab_* | wc -l

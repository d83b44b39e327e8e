#!/bin/bash
# A/B of the integer H(1) table (MTG_H1_INT) on the dimension-in-lane bodies: same harness, same flags, two table forms
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=14 -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-value -Wno-unused-result"
build() {  # name H K WS LS RS NP
  for v in 0 1; do
    hipcc $F -DMTG_H1_INT=$v -DLT_H=$2 -DLT_K=$3 -DLT_WS=$4 -DLT_LS=$5 -DLT_RS=$6 -DLT_NP=$7 ${EXTRA} dl_variant.hip -o bin/ab_$1_int$v &
  done
}
build n12k32 6 32 14 4 1 1
build n10k32 5 32 6 6 1 1
build n8k32 4 32 0 0 1 1
build n12k16 6 16 4 4 1 1
wait
build n10k16 5 16 0 0 0 1
build n12k8 6 8 0 0 0 2
build n10k8 5 8 0 0 0 2
build n8k8 4 8 0 0 0 2
wait
ls bin/ab_*

#!/bin/bash
# lab A/B of workspace / LDS step counts on the dimension-in-lane bodies (current lane code)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=14 -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-value -Wno-unused-result"
n=0
build() {  # name H K WS LS RS NP
  hipcc $F -DLT_H=$2 -DLT_K=$3 -DLT_WS=$4 -DLT_LS=$5 -DLT_RS=$6 -DLT_NP=$7 dl_variant.hip -o bin/ab_$1 2>/dev/null &
  n=$((n+1)); if [ $((n % 7)) = 0 ]; then wait; fi
}
build n12k16_ws4_ls4 6 16 4 4 1 1
build n12k16_ws1_ls1 6 16 1 1 1 1
build n12k16_ws2_ls2 6 16 2 2 1 1
build n12k16_ws3_ls3 6 16 3 3 1 1
build n10k32_ws6_ls6 5 32 6 6 1 1
build n10k32_ws4_ls4 5 32 4 4 1 1
build n10k32_ws5_ls5 5 32 5 5 1 1
build n12k32_ws14_ls4 6 32 14 4 1 1
build n12k32_ws11_ls4 6 32 11 4 1 1
build n12k32_ws12_ls4 6 32 12 4 1 1
build n12k32_ws13_ls4 6 32 13 4 1 1
build n12k24_ws10_ls4 6 24 10 4 1 1
build n12k24_ws5_ls4 6 24 5 4 1 1
build n12k24_ws7_ls4 6 24 7 4 1 1
build n10k24_ws2_ls2 5 24 2 2 1 1
build n10k24_ws0_ls0 5 24 0 0 1 1
wait
ls bin/ab_* | wc -l

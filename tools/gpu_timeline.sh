#!/bin/bash
# GPU visit: per-tile phase timeline of the long-chain dimension-in-lane kernels (tools/lab/long_timeline.hip).
# Build first (here, cross-compiled):  tools/gpu_timeline.sh build
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=14 -mllvm -pragma-unroll-threshold=1000000 -DMTG_LAB_TIMING -DMTG_LAB_TIMELINE=12"
cd "$(dirname "$0")/lab"
# (the hand-over variants bar / warm / early of profiles/r05_long_timeline.txt were built from commit 31baf6e)
declare -A V=( [base]="" )
declare -A S=( [12_32]="-DLT_H=6 -DLT_K=32 -DLT_WS=14 -DLT_LS=4 -DLT_RS=1" [10_32]="-DLT_H=5 -DLT_K=32 -DLT_WS=6 -DLT_LS=6 -DLT_RS=1"
               [12_16]="-DLT_H=6 -DLT_K=16 -DLT_WS=4 -DLT_LS=4 -DLT_RS=1" [8_32]="-DLT_H=4 -DLT_K=32 -DLT_WS=0 -DLT_LS=0 -DLT_RS=1" )
if [ "$1" = build ]; then
  for v in base; do
    for x in 12_32 10_32 12_16 8_32; do
      hipcc $F ${V[$v]} ${S[$x]} -Rpass-analysis=kernel-resource-usage long_timeline.hip -o long_timeline_${x}_$v 2> /tmp/lt_${x}_$v.log &
    done
    wait
  done
  grep -H "ScratchSize" /tmp/lt_*.log | sed 's/.*lt_\(.*\).log.*ScratchSize/\1 scratch/'; exit 0
fi
mkdir -p ../../gpurun_out/timeline
for x in 12_32 10_32 12_16 8_32; do for v in base; do echo "== $x $v"; timeout 120 ./long_timeline_${x}_$v 100000; done; done 2>&1 | tee ../../gpurun_out/timeline/long_timeline.txt

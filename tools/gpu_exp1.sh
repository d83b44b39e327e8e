#!/bin/bash
# GPU visit: launch-geometry microbenchmark, B = 10k store ablation, in-kernel phase timeline (measurement build).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/exp1
mkdir -p $OUT
cd $R
timeout 300 tools/micro/launch_geometry > $OUT/launch_geometry.txt 2>&1
timeout 300 python tools/exp_b10k.py 10000 > $OUT/ablation.txt 2>&1
MTG_HIP_LIB=$R/mav_trajectory_generation_amd/csrc/libmtg_hip_timing.so timeout 300 python tools/timing_probe.py 10000 split > $OUT/probe_split.txt 2>&1
MTG_HIP_LIB=$R/mav_trajectory_generation_amd/csrc/libmtg_hip_timing.so timeout 300 python tools/timing_probe.py 10000 fused > $OUT/probe_fused.txt 2>&1
cat $OUT/launch_geometry.txt $OUT/ablation.txt $OUT/probe_split.txt $OUT/probe_fused.txt

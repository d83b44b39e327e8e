#!/bin/bash
# Round 6, visit d: MTG_FLAG_REFINE on the device (tests, cost), the rank-deficient routes again
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06d; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_refine.py tests/test_pivot_threshold.py tests/test_gpu_parity.py tests/test_gpu_dimlane.py -m gpu -q -n 4 > $OUT/pytest_new.txt 2>&1; tail -n 40 $OUT/pytest_new.txt | cut -c1-600
timeout 300 python tools/bench_refine.py > $OUT/refine_cost.jsonl 2>&1; cat $OUT/refine_cost.jsonl | tail -12

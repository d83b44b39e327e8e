#!/bin/bash
# Round 4, visit D: cooperative form with prefetched inputs (tests with truth arbitration, timing on a finer small-batch grid),
# the compute-bound rows' VALU counters (FP64-issue roofline), bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04d; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_coop.py tests/test_bench.py -m gpu -q -n 6 > $OUT/pytest_first.txt 2>&1; tail -n 12 $OUT/pytest_first.txt | cut -c1-800
BATCHES=16,64,256,512,1024,1536,2048,2500,4000 timeout 600 python tools/bench_coop.py > $OUT/coop_vs_default.jsonl 2> $OUT/coop.err; cut -c1-200 $OUT/coop_vs_default.jsonl; tail -n 3 $OUT/coop.err
bash tools/gpu_profile_rows.sh r04 2>&1 | tail -n 2

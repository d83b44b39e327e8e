#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06f; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_refine.py tests/test_bench.py -m gpu -q -n 2 > $OUT/pytest_new.txt 2>&1; grep -n "^E  \|passed\|failed" $OUT/pytest_new.txt | cut -c1-700 | head -40

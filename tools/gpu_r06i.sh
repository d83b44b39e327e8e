#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06i; mkdir -p $OUT
cd $R
rm -f gpurun_out/parity_arbitration.jsonl
timeout 900 python -m pytest tests/test_bench.py tests/test_gpu_vs_reference.py tests/test_gpu_default_dispatch.py -m gpu -q -n 4 > $OUT/pytest_new.txt 2>&1; grep -n "^E  .*assert\|passed\|failed" $OUT/pytest_new.txt | cut -c1-300 | head
cp gpurun_out/parity_arbitration.jsonl $OUT/ 2>/dev/null; wc -l $OUT/parity_arbitration.jsonl

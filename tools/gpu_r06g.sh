#!/bin/bash
# Round 6, visit g: MTG_FLAG_REFINE with contraction off in the double-double code; run-time-K body with the head rows requested two steps ahead
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06g; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_refine.py tests/test_gpu_dimlane.py tests/test_gpu_default_dispatch.py tests/test_gpu_forms_fuzz.py -m gpu -q -n 4 > $OUT/pytest_new.txt 2>&1; grep -n "^E  .*assert\|passed\|failed" $OUT/pytest_new.txt | cut -c1-300 | head -30
for n in 8 10 12; do KS=36,40,50,64,100 MAXKB=10000000 python tools/bench_other_k.py $n 2>&1 | grep "^{" >> $OUT/other_k.jsonl; done
grep 100000 $OUT/other_k.jsonl
timeout 300 python tools/bench_refine.py > $OUT/refine_cost.jsonl 2>&1; grep "^{" $OUT/refine_cost.jsonl

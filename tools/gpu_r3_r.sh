#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03r; mkdir -p $OUT
cd $R
export SHAPES="12,32,5,3,1;8,32,3,3,1;10,32,4,3,1" BATCHES="10000,50000"
timeout 600 python tools/bench_outputs_matrix.py > $OUT/k32_dl.jsonl 2> $OUT/err.txt; cat $OUT/k32_dl.jsonl
MTG_NO_DL_EXTRA=1 timeout 600 python tools/bench_outputs_matrix.py > $OUT/k32_old.jsonl 2>> $OUT/err.txt; cat $OUT/k32_old.jsonl

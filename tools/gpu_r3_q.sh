#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03q; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_dimlane.py tests/test_gpu_parity.py tests/test_gpu_vs_reference.py tests/test_gpu_sequence.py -m gpu -x -q -n 4 > $OUT/pytest_extra.txt 2>&1; tail -5 $OUT/pytest_extra.txt
timeout 600 python tools/bench_outputs_matrix.py > $OUT/outputs_matrix_dl.jsonl 2> $OUT/outputs_matrix.err; cat $OUT/outputs_matrix_dl.jsonl; tail -3 $OUT/outputs_matrix.err

#!/bin/bash
# update path with whole-sector output: parity sweep + timing against the staging kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03n; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -m gpu -x -q -k "update or free" > $OUT/pytest_update.txt 2>&1; tail -5 $OUT/pytest_update.txt
timeout 300 python tools/bench_update.py > $OUT/update_slab.txt 2>&1
MTG_NO_SLAB=1 timeout 300 python tools/bench_update.py > $OUT/update_staging.txt 2>&1
paste -d'\n' $OUT/update_slab.txt $OUT/update_staging.txt

#!/bin/bash
# One GPU-box visit: selected parity tests + bench line.  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest $*" | tee $OUT/pytest.log
timeout 1500 python -m pytest "$@" -x -q 2>&1 | tail -40 | tee -a $OUT/pytest.log

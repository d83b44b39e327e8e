#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu" | tee $OUT/pytest.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee -a $OUT/pytest.log
echo "== bench" 
timeout 600 python bench.py --extra > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; cat $OUT/bench.json
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/rocprof_bench.json 2> $OUT/rocprof.err
tail -2 $OUT/rocprof.err
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"

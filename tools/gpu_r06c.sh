#!/bin/bash
# Round 6, visit c: the pruned build -- full GPU suite, per-shape kernel times (no regression expected: the removed code was compiled out)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06c; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -n 4 > $OUT/pytest_gpu.txt 2>&1; tail -n 15 $OUT/pytest_gpu.txt | cut -c1-400
python tools/bench_configs.py 2>&1 | grep "^{" > $OUT/configs.jsonl
python tools/bench_configs.py long 2>&1 | grep "^{" >> $OUT/configs.jsonl
grep -h "large\|long\|config5\|config2" $OUT/configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['N'], d['K'], d['B'], d['kernel_us'], round(d['frac_8TBps'], 3))"

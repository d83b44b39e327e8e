#!/bin/bash
# Round 6, visit a: the workspace accesses as global (not flat) loads / stores -- full GPU suite, per-shape kernel times, the
# chain lengths of the reference's own benchmark (K = 50 / 100), and the A/B harness on the same box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06a; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -n 4 -x > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt | cut -c1-300
python tools/bench_configs.py 2>&1 | grep "^{" > $OUT/configs.jsonl
python tools/bench_configs.py long 2>&1 | grep "^{" >> $OUT/configs.jsonl
for n in 8 10 12; do KS=17,24,31,50,100 MAXKB=10000000 python tools/bench_other_k.py $n 2>&1 | grep "^{" >> $OUT/other_k.jsonl; done
for v in /tmp/none $R/tools/lab/bin/dlv_*; do [ -x $v ] && $v 100000 $(basename $v) >> $OUT/variants.jsonl 2>&1; done
grep -h "large\|long" $OUT/configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['N'], d['K'], d['B'], d['kernel_us'], round(d['frac_8TBps'], 3))"
cat $OUT/other_k.jsonl | grep 100000
cat $OUT/variants.jsonl
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench_driver_args.json

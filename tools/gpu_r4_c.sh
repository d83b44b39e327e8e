#!/bin/bash
# Round 4, visit C: the row-cooperative kernel on hardware (tests, timing against the default forms), extrema after the
# stopping-rule / fast-reciprocal changes, the bench test fixed in visit B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_coop.py tests/test_extrema.py tests/test_bench.py -m gpu -q -n 6 > $OUT/pytest_first.txt 2>&1; tail -n 25 $OUT/pytest_first.txt | cut -c1-1200
timeout 600 python tools/bench_coop.py > $OUT/coop_vs_default.jsonl 2> $OUT/coop.err; cut -c1-200 $OUT/coop_vs_default.jsonl; tail -n 3 $OUT/coop.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> $OUT/bench.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('frac %.3f'%d['roofline']['frac'], 'next', {k: round(v['us'],1) for k,v in d['extra']['next'].items()})"

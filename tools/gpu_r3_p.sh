#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03p; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_dimlane.py tests/test_gpu_sequence.py -m gpu -x -q -n 4 > $OUT/pytest_dimlane.txt 2>&1; tail -5 $OUT/pytest_dimlane.txt
for i in 1; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/bench.err; python - $i <<'PY'
import json,sys
r=json.load(open('gpurun_out/r03p/bench%s.json'%sys.argv[1])); print(r['value'], r['roofline']['frac'], r['roofline']['kernel_us'], r['extra']['one_launch_per_batch']['us_per_step'], r['extra']['resident_buffers']['us_per_step'], r['extra']['rotating_buffers_96_steps_one_full_launch']['us_per_step'])
PY
done

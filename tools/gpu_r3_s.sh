#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03s; mkdir -p $OUT
cd $R
for v in ext ext ext rec rec; do
  if [ $v = rec ]; then export MTG_NO_EXT_EVENTS=1; else unset MTG_NO_EXT_EVENTS; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$v', r['value'], r['timed_region_wall_us'], r['roofline']['kernel_us'], r['roofline']['frac'])"
done
unset MTG_NO_EXT_EVENTS
python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('config5', r['value'], r['timed_region_wall_us'], r['roofline']['kernel_us'], r['roofline']['frac'])"
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('200 steps', r['value'], r['timed_region_wall_us'], r['roofline']['kernel_us'], r['roofline']['frac'])"
timeout 600 python -m pytest tests/test_gpu_sequence.py tests/test_gpu_parity.py -m gpu -x -q -k "sequence or events" 2>&1 | tail -2

#!/bin/bash
# does the timed 20-step launch of a FRESH process depend on how long the GPU has been busy before it?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$*', 'kernel_us %.1f' % r['roofline']['kernel_us'], 'frac %.3f' % r['roofline']['frac'], 'wall %.1f' % r['timed_region_wall_us']['wall'])"; }
for rep in 1 2 3; do
  sleep 15; one
  sleep 15; one --settle-ms 50
  sleep 15; one --settle-ms 300
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4

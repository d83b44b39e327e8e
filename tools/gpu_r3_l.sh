#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for g in 8 4 2 1; do
  echo "== MTG_DL_GRID_PER_CU=$g"
  MTG_DL_GRID_PER_CU=$g python tools/bench_configs.py 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['B']>=100000 and d['K']<=16: print(f\"{d['config']:14s} N={d['N']:2d} K={d['K']:2d} D={d['D']} B={d['B']:6d} {d['kernel_us']:8.2f} us  frac {d['frac_8TBps']:.3f}\")"
  MTG_DL_GRID_PER_CU=$g python tools/bench_configs.py long 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['K']==16: print(f\"long N={d['N']:2d} K={d['K']:2d} B={d['B']:6d} {d['kernel_us']:8.2f} us  frac {d['frac_8TBps']:.3f}\")"
done

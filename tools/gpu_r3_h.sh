#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03h; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_dimlane.py -m gpu -x -q -k "runtime_k" > $OUT/pytest_rt.txt 2>&1; tail -n 15 $OUT/pytest_rt.txt
tests/cpp/test_veneer > $OUT/test_veneer.txt 2>&1; tail -n 2 $OUT/test_veneer.txt
for rt in 0 1; do
  echo "== MTG_DL_RT=$rt"
  for n in 8 10 12; do MTG_DL_RT=$rt python tools/bench_other_k.py $n 2>&1 | grep "^{" | python -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin]
print('N=$n', ' '.join(f\"K{r['K']}/{r['B']//1000}k:{r['kernel_us']}({r['frac_8TBps']})\" for r in rows if r['K'] in (8,12,16,17,20,24,27,31,50,100) or r['K']<=5))"; done
  MTG_DL_RT=$rt python tools/bench_configs.py long 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(f\"long N={d['N']:2d} K={d['K']:2d} B={d['B']:6d} {d['kernel_us']:8.2f} us  frac {d['frac_8TBps']:.3f}\")"
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03k; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_dimlane.py -m gpu -x -q -k "runtime_k" > $OUT/pytest_rt.txt 2>&1; tail -n 4 $OUT/pytest_rt.txt
for rt in 1; do
  echo "== MTG_DL_RT=$rt"
  for n in 8 10 12; do KS=12,16,17,20,24,27,31,32,40,50,64,100 MAXKB=20000000 MTG_DL_RT=$rt python tools/bench_other_k.py $n 2>&1 | grep "^{" | python -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin]
print('N=$n', ' '.join(f\"K{r['K']}/{r['B']//1000}k:{r['kernel_us']}({r['frac_8TBps']})\" for r in rows))"; done
done

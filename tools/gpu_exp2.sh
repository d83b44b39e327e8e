#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/exp2
mkdir -p $OUT
cd $R
for v in 0 1; do
  echo "HIP_FORCE_DEV_KERNARG=$v" >> $OUT/kernarg.txt
  HIP_FORCE_DEV_KERNARG=$v timeout 300 python tools/exp_b10k.py 10000 >> $OUT/kernarg.txt 2>&1
done
echo "unset" >> $OUT/kernarg.txt
timeout 300 python tools/exp_b10k.py 10000 >> $OUT/kernarg.txt 2>&1
cat $OUT/kernarg.txt

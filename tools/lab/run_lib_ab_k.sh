#!/bin/bash
# old ($1: a library under tools/lab/bin) against the in-tree library on chain lengths given as "N:K,K,..." arguments
mkdir -p gpurun_out/r06k
old=$PWD/tools/lab/bin/$1; shift
for rep in 1 2; do
for spec in "$@"; do
  n=${spec%%:*}; ks=${spec##*:}
  for lib in old new; do
    if [ $lib = old ]; then export MTG_HIP_LIB=$old; else unset MTG_HIP_LIB; fi
    KS=$ks MAXKB=10000000 python tools/bench_other_k.py $n 2>&1 | grep '"B": 100000' | sed "s/^{/{\"lib\": \"$lib\", /"
  done
done
done | tee gpurun_out/r06k/lib_ab_k.jsonl | python3 -c "
import json,sys
d={}
for l in sys.stdin:
    r=json.loads(l); d.setdefault((r['N'],r['K']),{}).setdefault(r['lib'],[]).append(r['kernel_us'])
for k in sorted(d): print(k,d[k])"

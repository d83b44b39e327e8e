#!/bin/bash
# product-path A/B: tools/bench_configs.py with the round's previous library (tools/lab/bin/libmtg_hip_r06z.so) and the current one
mkdir -p gpurun_out/r06k
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export MTG_HIP_LIB=$PWD/tools/lab/bin/libmtg_hip_r06z.so; else unset MTG_HIP_LIB; fi
  python tools/bench_configs.py long 2>&1 | grep config | sed "s/^/{\"lib\": \"$lib\", /; s/{\"config/\"config/" 
  python tools/bench_configs.py 2>&1 | grep -E "config4-large|config2" | sed "s/^/{\"lib\": \"$lib\", /; s/{\"config/\"config/"
done
done > gpurun_out/r06k/${1:-lib_ab}.jsonl
python3 - gpurun_out/r06k/${1:-lib_ab}.jsonl <<'PY'
import json,sys
rows=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')]
d={}
for r in rows: d.setdefault((r['config'],r['N'],r['K'],r['B']),{}).setdefault(r['lib'],[]).append(r['kernel_us'])
for k in sorted(d): print(k, d[k])
PY

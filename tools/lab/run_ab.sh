#!/bin/bash
# runs every tools/lab/bin/ab_* twice (interleaved) at B = 100k; output: gpurun_out/r06k/ab.jsonl
mkdir -p gpurun_out/r06k
out=$PWD/gpurun_out/r06k/${1:-ab}.jsonl
: > $out
cd tools/lab/bin
for rep in 1 2; do
  for x in ab_*; do
    timeout 120 ./$x 100000 $x >> $out 2>&1
  done
done
python3 - "$out" <<'PY'
import json,sys
rows=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')]
best={}
for r in rows: best.setdefault(r['tag'],[]).append(r)
for t in sorted(best):
    rs=best[t]
    print(t, 'us_mean', [r['us_mean'] for r in rs], 'mhz', [r['shader_mhz'] for r in rs], 'frac', rs[0]['frac_8TBps'], 'vgprs', rs[0]['vgprs'], 'scratch', rs[0]['scratch'], rs[0]['hash'], rs[0]['status'], rs[0]['finite'])
PY

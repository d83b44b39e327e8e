#!/bin/bash
# Round 6, visit h: lab variants of the dimension-in-lane bodies with fewer register-resident steps (more LDS steps), and the same under
# launch bounds that hold the register allocation to 256 (two waves per SIMD)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06h; mkdir -p $OUT
cd $R
for v in $R/tools/lab/bin/dlv_*; do timeout 120 $v 100000 $(basename $v | sed 's/dlv_//') >> $OUT/variants.jsonl 2>&1; done
python - $OUT/variants.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print("%-28s occ %d wg %4d lds %6d vgprs %3d scratch %4d  %8.2f us (best %8.2f)  frac %.3f  %s %s" % (d["tag"], d["occ"], d["wg"], d["lds"], d["vgprs"], d["scratch"], d["us_mean"], d["us_best"], d["frac_8TBps"], d["hash"], "" if d["finite"] and d["status"] == 0 else "BAD"))
    else:
        print(l.strip()[:200])
PY

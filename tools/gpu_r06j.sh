#!/bin/bash
# Round 6, visit j: does a lone wave of the long-chain bodies get faster when the other SIMDs of its CU are idle?  (fewer persistent
# workgroups: 512 = every SIMD busy, 256 = two SIMDs per CU -- if the hardware places them so --, 128, 64)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06j; mkdir -p $OUT
cd $R
for v in n12k32 n10k32 n12k8_np1 n10k16 n10k8_np1; do for g in 512 384 256 128 64; do timeout 120 $R/tools/lab/bin/dlv_$v 100000 ${v}_wg$g $g >> $OUT/grid.jsonl 2>&1; done; done
python - $OUT/grid.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); tiles = (d["B"] + 20) // 21
        t = d["us_mean"] * d["wg"] / tiles
        print("%-18s wg %4d  %9.2f us  shader clock %5.0f MHz  one tile on one workgroup: %7.3f us = %8.0f cycles" % (d["tag"], d["wg"], d["us_mean"], d["shader_mhz"], t, t * d["shader_mhz"]))
PY

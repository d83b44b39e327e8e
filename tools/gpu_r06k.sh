#!/bin/bash
# Round 6, visit k: is the load-dependent slow-down of a tile CU-local or chip-wide?  512 waves as 256 two-wave workgroups (two busy SIMDs on
# every CU) against 128 four-wave workgroups (four busy SIMDs on half the CUs); and 1024 waves both ways.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06k; mkdir -p $OUT
cd $R
for v in n12k8 n10k8; do
  for g in 512 256 128 64; do timeout 120 $R/tools/lab/bin/dlv_${v}_np1 100000 ${v}_np1_wg$g $g >> $OUT/grid.jsonl 2>&1; done
  for g in 256 128 64 32; do timeout 120 $R/tools/lab/bin/dlv_${v}_np2 100000 ${v}_np2_wg$g $g >> $OUT/grid.jsonl 2>&1; done
done
python - $OUT/grid.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); tiles = (d["B"] + 20) // 21; units = (tiles + d["NP"] - 1) // d["NP"]
        t = d["us_mean"] * d["wg"] / units
        print("%-18s NP %d wg %4d waves %4d  %9.2f us  clock %5.0f MHz  one unit on one workgroup: %7.3f us = %8.0f cycles" % (d["tag"], d["NP"], d["wg"], d["wg"] * 2 * d["NP"], d["us_mean"], d["shader_mhz"], t, t * d["shader_mhz"]))
PY

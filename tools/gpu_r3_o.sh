#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03o; mkdir -p $OUT
cd $R
timeout 600 python tools/bench_layouts.py > $OUT/layouts.jsonl 2> $OUT/layouts.err; cat $OUT/layouts.jsonl; tail -3 $OUT/layouts.err

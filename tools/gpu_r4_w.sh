#!/bin/bash
# Round 4, visit W: the rebuilt final tree (MTG_PARTIAL_ALL off again): full suite, smoke, driver-args line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04w; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests -m gpu -q -n 6 -x > $OUT/pytest_gpu.txt 2>&1; tail -n 4 $OUT/pytest_gpu.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
python - $OUT/bench_driver_args.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
print("value %.4g cold %.4g dev us/step %.2f frac %.3f parity %s" % (d["value"], d.get("value_cold") or 0, r["device_us_per_step"], r["frac"], d["parity"]["ok"]))
PY

#!/bin/bash
# Final check of the round: full GPU suite, smoke, the driver's bench line, the N = 2 line on one GPU.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03final; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
python bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_two_ranks_one_gpu.json
python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.json 2>> $OUT/bench.err
for f in bench_driver_args bench_two_ranks_one_gpu bench_config4; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], "value %.3g" % d["value"], "dev us/step %.2f" % r["device_us_per_step"], "frac %.3f" % r["frac"], "keys", sorted(d.keys()))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done

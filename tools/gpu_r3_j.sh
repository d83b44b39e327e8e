#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03j; mkdir -p $OUT
cd $R
python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.json 2> $OUT/bench.err
python bench.py --config 4 --steps 20 --warmup 5 --sequence launches --no-extras > $OUT/bench_config4_launches.json 2>> $OUT/bench.err
python bench.py --config 4 --steps 96 --warmup 16 --no-extras > $OUT/bench_config4_96.json 2>> $OUT/bench.err
for f in bench_config4 bench_config4_launches bench_config4_96; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], "value %.3g" % d["value"], "dev us/step %.2f" % r["device_us_per_step"], "frac %.3f" % r["frac"], "launches", r["launches"],
          {k: round(v.get("us_per_step", 0), 2) for k, v in d.get("extra", {}).items() if isinstance(v, dict)})
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace(sys.argv[1].split("/")[-1], "bench.err")).read()[-1500:])
PY
done
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -n 4 $OUT/pytest_gpu.txt

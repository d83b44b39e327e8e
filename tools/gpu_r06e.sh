#!/bin/bash
# Round 6, visit e: refine tests again, the bench line with roofline.traffic measured by the invocation itself, FP64 counters of the (f) rows
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06e; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_refine.py tests/test_bench.py -m gpu -q -n 2 > $OUT/pytest_new.txt 2>&1; tail -n 12 $OUT/pytest_new.txt | cut -c1-400
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err ) 2>&1 | tail -3
python - $OUT/bench_driver_args.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value %.4g frac %.3f traffic/alg %s" % (d["value"], r["frac"], r.get("traffic_over_algorithmic")))
print(json.dumps(r.get("traffic_from_profile"))[:1200])
print("sustained", json.dumps(d.get("sustained"))[:400])
print("next", json.dumps({k: {kk: v.get(kk) for kk in ("us", "roofline")} for k, v in (d.get("extra", {}).get("next") or {}).items()})[:1500])
PY
bash tools/gpu_profile_rows.sh r06 > $OUT/profile_rows.txt 2>&1; tail -3 $OUT/profile_rows.txt | cut -c1-600
cp $R/gpurun_out/r06_next_rows_pmc.json $OUT/ 2>/dev/null

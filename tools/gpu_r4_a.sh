#!/bin/bash
# Round 4, visit A: full GPU suite on the refactored library (options instead of getenv, solve_impl split, basic solution behind
# the ABI, parity in the bench line), the bench lines with `parity`, the DPP row-broadcast microbenchmark, the phase-stagger sweep.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04a; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -n 6 > $OUT/pytest_gpu.txt 2>&1; tail -n 5 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.err
python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.json 2>> $OUT/bench.err
python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config5.json 2>> $OUT/bench.err
for f in bench_driver_args bench_config4 bench_config5; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], "value %.4g" % d["value"], "cold %.4g" % (d.get("value_cold") or 0), "frac %.3f" % r["frac"], "traffic", r.get("traffic_over_algorithmic"),
          "other", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (d.get("value_other_form") or {}).items() if k in ("value", "roofline_frac")})
    print("   parity", json.dumps(d.get("parity"))[:1500])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tools/micro/dpp_row_bcast.bin > $OUT/dpp_row_bcast.txt 2>&1; cat $OUT/dpp_row_bcast.txt
timeout 900 python tools/stagger_sweep.py > $OUT/stagger_sweep.jsonl 2> $OUT/stagger.err; cat $OUT/stagger_sweep.jsonl; tail -n 3 $OUT/stagger.err

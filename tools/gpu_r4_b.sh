#!/bin/bash
# Round 4, visit B: the bench test that failed in visit A (with its parity object printed), the rest of the suite, extrema /
# time-scaling rewrite (tests + timing), bench line after the settle-gap fix, DPP semantics with wait states, default stagger.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04b; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_bench.py tests/test_extrema.py tests/test_basic_solution.py -m gpu -q -n 4 > $OUT/pytest_first.txt 2>&1; tail -n 30 $OUT/pytest_first.txt | cut -c1-1500
timeout 1500 python -m pytest tests -m gpu -q -n 6 --deselect tests/test_bench.py > $OUT/pytest_gpu.txt 2>&1; tail -n 8 $OUT/pytest_gpu.txt | cut -c1-600
tools/micro/dpp_row_bcast.bin 2>&1 | head -3 > $OUT/dpp_row_bcast.txt; cat $OUT/dpp_row_bcast.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.err
python - $OUT/bench_driver_args.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value %.4g cold %.4g frac %.3f dev us/step %.2f" % (d["value"], d.get("value_cold") or 0, r["frac"], r["device_us_per_step"]), "other", d["value_other_form"]["value"], d["value_other_form"]["roofline_frac"])
print("parity", d["parity"]["ok"], d["parity"]["max_rel_err_vs_port"], d["parity"]["max_rel_err_vs_reference_build"])
print("next", {k: round(v["us"], 1) for k, v in d["extra"]["next"].items()})
PY
python tools/bench_configs.py long 2>&1 | grep "^{" > $OUT/configs_long.jsonl; cut -c1-220 $OUT/configs_long.jsonl

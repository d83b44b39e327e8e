#!/bin/bash
# Round 4, visit V: partial elimination in EVERY kernel's forward step (MTG_PARTIAL_ALL): full GPU suite, the bench lines, kernel times.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04v; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -n 6 > $OUT/pytest_gpu.txt 2>&1; tail -n 4 $OUT/pytest_gpu.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver_args_2.json 2>> $OUT/bench.err
python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_config5.json 2>> $OUT/bench.err
python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_config3.json 2>> $OUT/bench.err
python tools/bench_configs.py 2>&1 | grep "^{" > $OUT/configs.jsonl
python tools/bench_configs.py long 2>&1 | grep "^{" >> $OUT/configs.jsonl
for f in bench_driver_args bench_driver_args_2 bench_config3 bench_config5; do
  python - $OUT/$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
print(sys.argv[1].split("/")[-1], "value %.4g cold %.4g dev us/step %.2f frac %.3f parity %s %s other-form us %s" % (d["value"], d.get("value_cold") or 0, r["device_us_per_step"], r["frac"], d["parity"]["ok"], d["parity"].get("max_rel_err_vs_reference_build"), (d.get("value_other_form") or {}).get("device_us_per_step")))
PY
done
python - $OUT/configs.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["config"], "N", d["N"], "K", d["K"], "D", d["D"], "B", d["B"], "kernel_us %.2f frac %.3f" % (d["kernel_us"], d["frac_8TBps"]))
PY

#!/bin/bash
# Round 4, visit U (last): full GPU suite on the final tree, config 4's profile and line again (step distribution retuned), driver-args line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04u; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -q -n 6 > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
bash tools/gpu_profile3.sh r04_config4 --config 4 --steps 20 --warmup 5 > $OUT/profile_config4.json 2> $OUT/profile.err
cp $R/gpurun_out/r04_config4_pmc_traffic.json $R/gpurun_out/r04_config4_kernel_stats.csv $R/gpurun_out/r04_config4_kernel_trace_solve_launches.csv $OUT/ 2>/dev/null
python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err
python tools/bench_configs.py 2>&1 | grep "^{" > $OUT/configs.jsonl
python tools/bench_configs.py long 2>&1 | grep "^{" >> $OUT/configs.jsonl
for f in bench_driver_args bench_config4; do
  python - $OUT/$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
print(sys.argv[1].split("/")[-1], "value %.4g cold %.4g dev us/step %.2f frac %.3f traffic/alg %s parity %s" % (d["value"], d.get("value_cold") or 0, r["device_us_per_step"], r["frac"], r.get("traffic_over_algorithmic"), d["parity"]["ok"]))
PY
done

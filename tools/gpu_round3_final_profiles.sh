#!/bin/bash
# Round-3 evidence run on the FINAL code of the round (tag r03z): full GPU suite, the bench lines (driver's arguments, other configs, latency form, two ranks on the one
# GPU), rocprofv3 kernel trace / stats + calibrated PMC traffic of the bench command, the next-step kernels' profiles, per-config
# kernel times, the mixed request.  Outputs under gpurun_out/r03/ (copied into profiles/ by hand).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03z; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -n 4 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-next --extra > $OUT/bench_200_steps.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-next --sequence launches > $OUT/bench_launches.json 2>> $OUT/bench.err
python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config3.json 2>> $OUT/bench.err
python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.json 2>> $OUT/bench.err
python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config5.json 2>> $OUT/bench.err
python bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_two_ranks_one_gpu.json
bash tools/gpu_profile3.sh r03z_driver_args > $OUT/profile_driver_args.json 2> $OUT/profile.err
bash tools/gpu_profile3.sh r03z_steps96 --steps 96 --warmup 96 > $OUT/profile_steps96.json 2>> $OUT/profile.err
bash tools/gpu_profile3.sh r03z_config4 --config 4 --steps 20 --warmup 5 > $OUT/profile_config4.json 2>> $OUT/profile.err
bash tools/gpu_profile_next.sh r03z > $OUT/profile_next.txt 2>&1
cp $R/gpurun_out/r03z_*_pmc_traffic.json $R/gpurun_out/r03z_*_kernel_stats.csv $R/gpurun_out/r03z_*_kernel_trace_solve_launches.csv $R/gpurun_out/r03z_next_pmc.json $OUT/ 2>/dev/null
python tools/bench_configs.py 2>&1 | grep "^{" > $OUT/configs.jsonl
python tools/bench_configs.py long 2>&1 | grep "^{" >> $OUT/configs.jsonl
(for n in 10 8 12; do MAXKB=20000000 python tools/bench_other_k.py $n 2>&1 | grep "^{"; done) > $OUT/other_chain_lengths.jsonl
python tools/bench_mixed.py 2500 merged 2>&1 | grep "^{" > $OUT/mixed_config4.jsonl
tools/cpp/polynomial_timing_evaluation > $OUT/veneer_timing_evaluation.txt 2>&1
tests/cpp/test_veneer > $OUT/test_veneer.txt 2>&1; tail -n 1 $OUT/test_veneer.txt
for f in bench_driver_args bench_200_steps bench_launches bench_config3 bench_config4 bench_config5 bench_two_ranks_one_gpu; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], "value %.3g" % d["value"], "dev us/step %.2f" % r["device_us_per_step"], "frac %.3f" % r["frac"],
          {k: round(v.get("us_per_step", v.get("kernel_us", 0)), 2) for k, v in d.get("extra", {}).items() if isinstance(v, dict) and ("us_per_step" in v or "kernel_us" in v)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
du -sh $R/gpurun_out

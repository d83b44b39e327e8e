#!/bin/bash
# Round-6 evidence run on the FINAL code of the round: full GPU suite, smoke, the bench lines of every config, rocprofv3 kernel
# trace / stats + calibrated PMC traffic of the bench command for configs 2 / 3 / 4 / 5 (-> r06_config<N>_pmc_traffic.json, which
# bench.py reads back as roofline.traffic), the compute-bound rows' VALU counters, per-shape kernel times.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06z; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -n 4 > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
timeout 300 python bench.py > $OUT/bench_defaults.json 2>> $OUT/bench.err
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config3.json 2>> $OUT/bench.err
timeout 400 python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.json 2>> $OUT/bench.err
timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config5.json 2>> $OUT/bench.err
timeout 300 python bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_two_ranks_one_gpu.json
timeout 300 python bench.py --gpus 1 --exercise-collectives --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" > $OUT/bench_one_rank_rccl.json
bash tools/gpu_profile3.sh r06_config2 --steps 20 --warmup 5 > $OUT/profile_config2.json 2> $OUT/profile.err
bash tools/gpu_profile3.sh r06_config4 --config 4 --steps 20 --warmup 5 > $OUT/profile_config4.json 2>> $OUT/profile.err
bash tools/gpu_profile3.sh r06_config5 --config 5 --steps 20 --warmup 5 > $OUT/profile_config5.json 2>> $OUT/profile.err
bash tools/gpu_profile3.sh r06_config3 --config 3 --steps 20 --warmup 5 > $OUT/profile_config3.json 2>> $OUT/profile.err
bash tools/gpu_profile_rows.sh r06 > $OUT/profile_rows.txt 2>&1
cp $R/gpurun_out/r06_config*_pmc_traffic.json $R/gpurun_out/r06_config*_kernel_stats.csv $R/gpurun_out/r06_config*_kernel_trace_solve_launches.csv $R/gpurun_out/r06_next_rows_pmc.json $OUT/ 2>/dev/null
python tools/bench_configs.py 2>&1 | grep "^{" > $OUT/configs.jsonl
python tools/bench_configs.py long 2>&1 | grep "^{" >> $OUT/configs.jsonl
for n in 8 10 12; do KS=17,24,31,50,100 MAXKB=10000000 python tools/bench_other_k.py $n 2>&1 | grep "^{" >> $OUT/other_k.jsonl; done
timeout 300 python tools/bench_refine.py 2>&1 | grep "^{" > $OUT/refine_cost.jsonl
timeout 200 python tools/bench_extrema.py > $OUT/extrema.txt 2>&1
timeout 120 tools/c/roundtrip 100000 8 > $OUT/roundtrip.txt 2>&1
for f in bench_driver_args bench_defaults bench_config3 bench_config4 bench_config5 bench_two_ranks_one_gpu bench_one_rank_rccl; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    p = d.get("parity") or {}
    su = d.get("sustained") or {}
    print(sys.argv[1].split("/")[-1], "sustained frac %s clock %s" % (su.get("roofline_frac"), su.get("shader_clock_mhz")), "aos", (d.get("value_aos_inputs") or {}).get("roofline_frac"),
          "traffic measured by", ((d["roofline"].get("traffic_from_profile") or {}).get("measured") or (d["roofline"].get("traffic_from_profile") or {}).get("file")))
    print(sys.argv[1].split("/")[-1], "value %.3g" % d["value"], "cold %.3g" % (d.get("value_cold") or 0), "dev us/step %.2f" % r["device_us_per_step"], "frac %.3f" % r["frac"],
          "sets", d["config"].get("buffer_sets"), "traffic/alg", r.get("traffic_over_algorithmic"), "parity", p.get("ok"), p.get("max_rel_err_vs_reference_build"),
          "next", {k: round(v["us"], 1) for k, v in (d.get("extra", {}).get("next") or {}).items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
du -sh $R/gpurun_out

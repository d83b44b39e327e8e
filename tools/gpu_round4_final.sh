#!/bin/bash
# Round-4 evidence run on the FINAL code of the round: full GPU suite, the bench lines of every config, rocprofv3 kernel trace /
# stats + calibrated PMC traffic of the bench command for configs 2 / 3 / 4 / 5 (-> profiles/r04_config<N>_pmc_traffic.json, which
# bench.py reads back as roofline.traffic), the compute-bound rows' VALU counters, per-config kernel times.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04z; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -n 6 > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-next --sequence launches > $OUT/bench_launches.json 2>> $OUT/bench.err
python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config3.json 2>> $OUT/bench.err
python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.json 2>> $OUT/bench.err
python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config5.json 2>> $OUT/bench.err
python bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_two_ranks_one_gpu.json
bash tools/gpu_profile3.sh r04_config2 --steps 20 --warmup 5 > $OUT/profile_config2.json 2> $OUT/profile.err
bash tools/gpu_profile3.sh r04_config3 --config 3 --steps 20 --warmup 5 > $OUT/profile_config3.json 2>> $OUT/profile.err
bash tools/gpu_profile3.sh r04_config4 --config 4 --steps 20 --warmup 5 > $OUT/profile_config4.json 2>> $OUT/profile.err
bash tools/gpu_profile3.sh r04_config5 --config 5 --steps 20 --warmup 5 > $OUT/profile_config5.json 2>> $OUT/profile.err
bash tools/gpu_profile_rows.sh r04 > $OUT/profile_rows.txt 2>&1
cp $R/gpurun_out/r04_config*_pmc_traffic.json $R/gpurun_out/r04_config*_kernel_stats.csv $R/gpurun_out/r04_config*_kernel_trace_solve_launches.csv $R/gpurun_out/r04_next_rows_pmc.json $OUT/ 2>/dev/null
python tools/bench_configs.py 2>&1 | grep "^{" > $OUT/configs.jsonl
python tools/bench_configs.py long 2>&1 | grep "^{" >> $OUT/configs.jsonl
(for n in 10 8 12; do KS=17,24,27,40,50,100 MAXKB=20000000 python tools/bench_other_k.py $n 2>&1 | grep "^{"; done) > $OUT/other_chain_lengths.jsonl
BATCHES=64,1024,2500 timeout 300 python tools/bench_coop.py > $OUT/coop_vs_default.jsonl 2>/dev/null
for f in bench_driver_args bench_launches bench_config3 bench_config4 bench_config5 bench_two_ranks_one_gpu; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    p = d.get("parity") or {}
    print(sys.argv[1].split("/")[-1], "value %.3g" % d["value"], "cold %.3g" % (d.get("value_cold") or 0), "dev us/step %.2f" % r["device_us_per_step"], "frac %.3f" % r["frac"],
          "traffic/alg", r.get("traffic_over_algorithmic"), "parity", p.get("ok"), p.get("max_rel_err_vs_reference_build"),
          "next", {k: round(v["us"], 1) for k, v in (d.get("extra", {}).get("next") or {}).items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
du -sh $R/gpurun_out

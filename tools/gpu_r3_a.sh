#!/bin/bash
# Round 3, first visit: the queue form of mtg_solve_linear_sequence -- parity tests, the driver's bench line, other configs,
# rocprof of the bench command.  Outputs under gpurun_out/r03a/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03a; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_sequence.py tests/test_bench.py -m gpu -x -q > $OUT/pytest_sequence.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_dimlane.py tests/test_gpu_parity.py -m gpu -x -q -k "own_status or cross_over or generator_and_compare or sequence_with_events or per_trajectory_status or graph_replays" > $OUT/pytest_misc.txt 2>&1
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver_args_2nd.json 2>> $OUT/bench_driver_args.err
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_200.json 2> $OUT/bench_200.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sequence launches > $OUT/bench_launches.json 2>> $OUT/bench_200.err
python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config3.json 2>> $OUT/bench_200.err
python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.json 2>> $OUT/bench_200.err
python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config5.json 2>> $OUT/bench_200.err
python bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_two_ranks_one_gpu.json
bash tools/gpu_profile3.sh r03a_driver_args > $OUT/profile_driver_args.json 2> $OUT/profile.err
bash tools/gpu_profile3.sh r03a_steps96 --steps 96 --warmup 96 > $OUT/profile_steps96.json 2>> $OUT/profile.err
cp $R/gpurun_out/r03a_*_pmc_traffic.json $R/gpurun_out/r03a_*_kernel_stats.csv $R/gpurun_out/r03a_*_kernel_trace_solve_launches.csv $OUT/ 2>/dev/null
tail -3 $OUT/pytest_sequence.txt $OUT/pytest_misc.txt
for f in bench_driver_args bench_driver_args_2nd bench_200 bench_launches bench_config3 bench_config4 bench_config5; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], "value %.3g" % d["value"], "ms/step %.4f" % d["ms_per_step"], "dev us/step %.2f" % r["device_us_per_step"],
          "frac %.3f" % r["frac"], {k: round(v.get("us_per_step", v.get("kernel_us", 0)), 2) for k, v in d.get("extra", {}).items() if isinstance(v, dict) and ("us_per_step" in v or "kernel_us" in v)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done

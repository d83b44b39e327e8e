#!/bin/bash
# round 5, visit B: the whole GPU suite on the relative-threshold / structural-rank build, the bench line with the new default
# rotation, per-shape kernel times (against profiles/r04z_configs.jsonl: what the threshold costs)
mkdir -p gpurun_out/r05b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05b/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05b/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05b/bench_driver_args.json 2> gpurun_out/r05b/bench_driver_args.err
timeout 300 python tools/bench_configs.py > gpurun_out/r05b/configs.jsonl 2> gpurun_out/r05b/configs.err
timeout 300 python tools/bench_configs.py long >> gpurun_out/r05b/configs.jsonl 2>> gpurun_out/r05b/configs.err
timeout 400 python bench.py --config 4 --steps 20 --warmup 5 --no-next > gpurun_out/r05b/bench_config4.json 2> gpurun_out/r05b/bench_config4.err
timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --no-next --no-cpu-baseline > gpurun_out/r05b/bench_config5.json 2> gpurun_out/r05b/bench_config5.err
tail -5 gpurun_out/r05b/tests.log

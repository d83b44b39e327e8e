#!/bin/bash
# round 5, visit A: the default-dispatch parity tests + is the config-2 / config-5 roofline HBM or Infinity Cache?
# (rotation over 16 / 48 / 128 buffer sets; VERDICT round 4, "What's weak" 6)
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_default_dispatch.py tests/test_coop.py -m gpu -x -q > gpurun_out/r05a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05a/tests.log
for s in 16 48 128; do
  timeout 300 python bench.py --steps 20 --warmup 5 --buffer-sets $s --no-next --no-cpu-baseline --no-parity > gpurun_out/r05a/bench_c2_sets$s.json 2> gpurun_out/r05a/bench_c2_sets$s.err
done
for s in 16 48 128; do
  timeout 300 python bench.py --steps 96 --warmup 5 --buffer-sets $s --no-next --no-cpu-baseline --no-parity --no-extras > gpurun_out/r05a/bench_c2_steps96_sets$s.json 2> gpurun_out/r05a/bench_c2_steps96_sets$s.err
done
for s in 16 48; do
  timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --buffer-sets $s --no-next --no-cpu-baseline --no-parity > gpurun_out/r05a/bench_c5_sets$s.json 2> gpurun_out/r05a/bench_c5_sets$s.err
done
tail -5 gpurun_out/r05a/tests.log

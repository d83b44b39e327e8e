#!/bin/bash
# round 5, visit E: the whole GPU suite on the reverted pivot test + shadow plans + communicator + reference's own tests
mkdir -p gpurun_out/r05e
timeout 1900 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r05e/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05e/tests.log
timeout 120 tools/c/roundtrip 100000 8 > gpurun_out/r05e/roundtrip.log 2>&1; echo "rc=$?" >> gpurun_out/r05e/roundtrip.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05e/bench_driver_args.json 2> gpurun_out/r05e/bench_driver_args.err
tail -25 gpurun_out/r05e/tests.log; cat gpurun_out/r05e/roundtrip.log

#!/bin/bash
# round 5, visit G: PC sampling attempt (instruction-level stall profile of N = 12 / K = 32), extrema after the partition-point
# change, the bench contract tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_extrema.py tests/test_bench.py "tests/test_gpu_vs_reference.py::test_extrema_and_time_scaling_vs_reference" tests/test_cpp_veneer.py tests/test_reference_own_tests.py -m gpu -q --maxfail=10 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
timeout 300 python tools/bench_extrema.py > $O/extrema.txt 2>&1
bash tools/gpu_pc_sampling.sh $O/pcs 12 32 100000 40 > $O/pcs_script.log 2>&1
tail -5 $O/tests.log | cut -c1-200; head -8 $O/extrema.txt; tail -30 $O/pcs_script.log

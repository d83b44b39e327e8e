#!/bin/bash
# round 5, visit D: the shadow-plan basic solution, the reference's own test binaries, the tests touched since visit C
mkdir -p gpurun_out/r05d
timeout 1500 python -m pytest tests/test_pivot_threshold.py tests/test_basic_solution.py tests/test_reference_own_tests.py tests/test_cpp_veneer.py \
   "tests/test_gpu_dimlane.py::test_dimlane_extra_outputs" tests/test_gpu_vs_reference.py -m gpu -q --maxfail=25 > gpurun_out/r05d/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05d/tests.log
for b in polynomial_optimization_veneer polynomial_optimization_refclass; do
  MTG_COMPAT_SINGLE_CALLS=host MTG_REF_TESTS_BACKEND=host timeout 600 tests/ref_tests/bin/$b '--gtest_filter=-*UnconstrainedNonlinear*:*.TimeScaling/*' > gpurun_out/r05d/$b.log 2>&1
  echo "rc=$?" >> gpurun_out/r05d/$b.log
done
tail -30 gpurun_out/r05d/tests.log

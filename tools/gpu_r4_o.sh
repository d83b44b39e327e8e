#!/bin/bash
# Round 4, visit O: factor store: bit-identity diagnostics, then the tests that compare kernel forms.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04o; mkdir -p $OUT; cd $R
python tools/compare_kernel_forms.py 2>/dev/null | tee $OUT/diag.txt | head -20
timeout 900 python -m pytest tests/test_gpu_dimlane.py tests/test_gpu_parity.py -m gpu -q -n 6 -k "extra_outputs or merged or mixed or default_for_the_bench or rt" > $OUT/pytest_forms.txt 2>&1; tail -n 8 $OUT/pytest_forms.txt | cut -c1-300

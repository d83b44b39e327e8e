#!/bin/bash
# A/B of library builds: bench line (10k) + big batches per variant, plus the FP64 microbenchmark.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R

for spec in "$@"; do
  lib=${spec%%:*}; dims=${spec##*:}; [ "$dims" = "$spec" ] && dims=auto
  echo "== $lib dims=$dims"
  MTG_HIP_LIB=$R/mav_trajectory_generation_amd/csrc/$lib timeout 300 python bench.py --extra --no-cpu-baseline --steps 100 --dims $dims 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  10k: %.1f us/step kernel %.2f us frac %.3f | 125k: %.1f us %.3f | 1M: %.1f us %.3f'%(d['ms_per_step']*1e3,d['roofline']['kernel_us'],d['roofline']['frac'],d['extra']['batch_125000']['kernel_us'],d['extra']['batch_125000']['frac_of_8TBps'],d['extra']['batch_1000000']['kernel_us'],d['extra']['batch_1000000']['frac_of_8TBps']))"
done

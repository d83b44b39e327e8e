#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQC?_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TA_[A-Z0-9_]+)\b" | sort -u > $OUT/counters.txt; wc -l $OUT/counters.txt
B=${1:-1000000}
pass() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$n -o p -- python $R/tools/prof_driver.py $B 3 > $OUT/pmc_$n.log 2>&1
  f=$(find $OUT/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "mtg_solve" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  %-32s mean/dispatch = %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
}
echo "== icache"; pass ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
echo "== sq4"; pass sq4 SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_FLAT
cd $R
for lib in libmtg_hip.so libmtg_hip_nostore.so; do
  echo "== $lib"
  MTG_HIP_LIB=$R/mav_trajectory_generation_amd/csrc/$lib timeout 300 python bench.py --extra --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  10k: %.1f us/step kernel %.2f us frac %.3f | 125k: %.1f us %.3f | 1M: %.1f us %.3f'%(d['ms_per_step']*1e3,d['roofline']['kernel_us'],d['roofline']['frac'],d['extra']['batch_125000']['kernel_us'],d['extra']['batch_125000']['frac_of_8TBps'],d['extra']['batch_1000000']['kernel_us'],d['extra']['batch_1000000']['frac_of_8TBps']))"
done

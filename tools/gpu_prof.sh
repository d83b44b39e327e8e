#!/bin/bash
# PMC passes on the solve kernel + FP64 microbenchmark.  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== microbench"; timeout 120 $R/tools/micro/fp64_issue | tee $OUT/fp64_issue.txt
B=${1:-1000000}
pass() { # name counters...
  n=$1; shift
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
    print("  %-28s mean/dispatch = %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
}
echo "== pmc sq1"; pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
echo "== pmc sq2"; pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
echo "== pmc sq3"; pass sq3 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU_MFMA_F64 GRBM_GUI_ACTIVE
echo "== pmc fetch"; pass fetch FETCH_SIZE
echo "== pmc write"; pass write WRITE_SIZE
echo "== pmc tcc"; pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum

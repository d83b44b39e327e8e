#!/bin/bash
# round 5, visit F: full GPU suite; config-4 bench with the request built inside the timed region; a PC-sampling attempt on the
# N = 12 / K = 32 long-chain kernel (instruction-level stall profile: what rounds 3-4 lacked)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout 1900 python -m pytest tests -m gpu -q --maxfail=25 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
timeout 400 python bench.py --config 4 --steps 20 --warmup 5 --no-next > $O/bench_config4.json 2> $O/bench_config4.err
cd /tmp && export TMPDIR=/tmp
for method in stochastic host_trap; do
  if [ $method = stochastic ]; then unit=cycles; interval=1048576; else unit=time; interval=100; fi
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval \
     --output-format csv -d $O/pcs_$method -o pcs -- python $R/tools/long_chain_driver.py 12 32 100000 40 > $O/pcs_$method.log 2>&1
  echo "rc=$?" >> $O/pcs_$method.log
  find $O/pcs_$method -name "*.csv" | head -5 >> $O/pcs_$method.log
  f=$(find $O/pcs_$method -name "*pc_sampling*csv" | head -1)
  if [ -n "$f" ]; then
    wc -l $f >> $O/pcs_$method.log; head -3 $f >> $O/pcs_$method.log
    python - "$f" > $O/pcs_${method}_summary.txt 2>&1 <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("samples", len(rows), "columns", list(rows[0].keys()) if rows else None)
def col(*names):
    for n in names:
        if rows and n in rows[0]: return n
    return None
ci, cs, ct, cw = col("Instruction"), col("Stall_Reason", "Stall_Reason_Not_Issued"), col("Instruction_Type"), col("Wave_Issued_Instruction", "Wave_Issued")
for c in (cs, ct, cw):
    if c:
        print("\n==", c)
        for k, v in collections.Counter(r[c] for r in rows).most_common(20): print(f"{v:8d} {k}")
if ci:
    print("\n== top instructions (opcode)")
    for k, v in collections.Counter(r[ci].split()[0] if r[ci] else "?" for r in rows).most_common(40): print(f"{v:8d} {k}")
    if cs:
        print("\n== opcode x stall")
        for k, v in collections.Counter(((r[ci].split()[0] if r[ci] else "?"), r[cs]) for r in rows).most_common(60): print(f"{v:8d} {k}")
PY
    # keep the raw file small: gzip, cap 20 MB
    gzip -c $f | head -c 20000000 > $O/pcs_${method}_samples.csv.gz
  fi
  rm -rf $O/pcs_$method
done
tail -8 $O/tests.log | cut -c1-200; cat $O/pcs_stochastic.log | tail -12

#!/bin/bash
# Instruction-level stall profile of a long-chain kernel with rocprofv3 PC sampling (beta): lists what the agent supports, then
# tries configurations until one is accepted; summarises samples per stall reason / instruction class.
# usage: gpu_pc_sampling.sh OUTDIR N K B ITERS
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=${1:-$R/gpurun_out/pcs}; N=${2:-12}; K=${3:-32}; B=${4:-100000}; IT=${5:-40}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --list-avail > $O/list_avail.txt 2>&1
grep -i -n -B2 -A14 "pc.sampl" $O/list_avail.txt | head -80 > $O/list_avail_pc_sampling.txt
ok=""
for cfg in "stochastic cycles 65536" "stochastic cycles 16384" "stochastic cycles 262144" "stochastic cycles 4096" "stochastic cycles 1024" \
           "host_trap time 1000" "host_trap time 100" "host_trap time 10" "host_trap time 1" "host_trap time 10000" "host_trap instructions 10000" "stochastic instructions 10000"; do
  set -- $cfg
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 \
     --output-format csv -d $O/run -o pcs -- python $R/tools/long_chain_driver.py $N $K $B $IT > $O/try_$1_$2_$3.log 2>&1
  rc=$?
  f=$(find $O/run -name "*pc_sampling*csv" 2>/dev/null | head -1)
  echo "$cfg rc=$rc file=$f" >> $O/attempts.txt
  if [ -n "$f" ] && [ $(wc -l < $f) -gt 100 ]; then ok="$cfg"; break; fi
  rm -rf $O/run
done
echo "accepted: $ok" >> $O/attempts.txt
if [ -n "$ok" ]; then
  f=$(find $O/run -name "*pc_sampling*csv" | head -1)
  head -3 $f > $O/sample_head.txt
  python - "$f" > $O/summary.txt 2>&1 <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("samples", len(rows)); print("columns", list(rows[0].keys()))
cols = list(rows[0].keys())
ci = next((c for c in cols if c.lower() == "instruction"), None)
for c in cols:
    vals = collections.Counter(r[c] for r in rows)
    if 1 < len(vals) <= 40 and c != ci:
        print("\n==", c)
        for k, v in vals.most_common(40): print(f"{v:8d} {100.0*v/len(rows):5.1f}% {k}")
if ci:
    op = lambda r: (r[ci].split()[0] if r[ci] else "?")
    print("\n== opcode")
    for k, v in collections.Counter(op(r) for r in rows).most_common(50): print(f"{v:8d} {100.0*v/len(rows):5.1f}% {k}")
    for c in cols:
        if "stall" in c.lower() or "issued" in c.lower():
            print("\n== opcode x", c)
            for k, v in collections.Counter((op(r), r[c]) for r in rows).most_common(80): print(f"{v:8d} {100.0*v/len(rows):5.1f}% {k}")
PY
  gzip -c $f | head -c 12000000 > $O/samples.csv.gz
fi
rm -rf $O/run
cat $O/attempts.txt; head -40 $O/list_avail_pc_sampling.txt

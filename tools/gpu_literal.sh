#!/bin/bash
# GPU visit: the MFMA evidence variant reachable through the library -- parity test, timing, MFMA counters.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/literal; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_literal_mfma.py tests/test_abi.py -q -x 2>&1 | tail -5 | tee $O/pytest_tail.txt
timeout 300 python tools/literal_mfma_driver.py > $O/timing.jsonl 2> $O/timing.err; cat $O/timing.jsonl; tail -2 $O/timing.err
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/mfma_counters_available.txt; cat $O/mfma_counters_available.txt; echo
for c in SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python tools/literal_mfma_driver.py 1048576 2 > /dev/null 2> $O/pmc_$c.err
done
python3 - <<'PY'
import csv, glob, collections, json, os
O = "gpurun_out/literal"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "literal_kernel" in r["Kernel_Name"] or "scaled_kernel" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(O + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY

#!/bin/bash
# Instruction fetch of the fully unrolled solve kernels (rocprofv3 --pmc, its own run): I-cache requests / hits / misses and
# the mean fetch latency, beside the wave-cycle breakdown.   usage: gpu_pmc_ifetch.sh TAG [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/${TAG}_if -o p -- python $R/bench.py --no-cpu-baseline --no-extras --settle-ms 0 "$@" > /dev/null 2> $OUT/${TAG}_if.err
timeout 600 rocprofv3 --kernel-trace --pmc SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVES --output-format csv -d $OUT/${TAG}_if2 -o p -- python $R/bench.py --no-cpu-baseline --no-extras --settle-ms 0 "$@" > /dev/null 2> $OUT/${TAG}_if2.err
python - $OUT $TAG <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for sub in ("_if", "_if2"):
    f = glob.glob(os.path.join(out, tag + sub, "**", "*counter_collection.csv"), recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "mtg_solve" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        for name, v in c.items():
            res[k][name] = sum(v[len(v) // 2:]) / len(v[len(v) // 2:])
json.dump(res, open(os.path.join(out, tag + "_ifetch.json"), "w"), indent=1)
for k, d in res.items():
    print(k); print("   ", {n: round(v, 1) for n, v in d.items()})
    if d.get("SQC_ICACHE_REQ"):
        print("    hit rate %.3f  misses/req %.3f  mean fetch latency (cycles) %.1f" % (d.get("SQC_ICACHE_HITS", 0) / d["SQC_ICACHE_REQ"], d.get("SQC_ICACHE_MISSES", 0) / d["SQC_ICACHE_REQ"], d.get("SQ_IFETCH_LEVEL", 0) / max(d.get("SQ_IFETCH", 1), 1)))
PY
rm -rf $OUT/${TAG}_if $OUT/${TAG}_if2

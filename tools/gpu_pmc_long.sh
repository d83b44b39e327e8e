#!/bin/bash
# FP64-issue evidence for the long-chain kernels: VALU instructions and busy cycles per launch (rocprofv3 --pmc, its own
# run, kernel trace only) next to the kernel durations -> gpurun_out/${TAG}_long_pmc.json
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_long_pmc -o p -- python $R/tools/bench_configs.py long > $OUT/${TAG}_long_pmc_bench.txt 2> $OUT/${TAG}_long_pmc.err
python - $OUT $TAG <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(out, tag + "_long_pmc", "**", "*counter_collection.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "mtg_solve_dl" in r["Kernel_Name"]:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tr = glob.glob(os.path.join(out, tag + "_long_pmc", "**", "*kernel_trace.csv"), recursive=True)
dur = collections.defaultdict(list)
if tr:
    for r in csv.DictReader(open(tr[0])):
        if "mtg_solve_dl" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = []
for k, c in acc.items():
    d = {"kernel": k, "launches": len(dur[k]), "mean_ns_profiled": sum(dur[k]) / max(1, len(dur[k]))}
    for name, v in c.items():
        d[name] = sum(v) / len(v)
    if "SQ_INSTS_VALU" in d and d["mean_ns_profiled"] > 0:
        # one VALU wave-instruction occupies its SIMD's FP64 pipe for 4 cycles (16 lanes / clk); 1024 SIMDs
        d["valu_issue_cycles_per_simd"] = d["SQ_INSTS_VALU"] * 4 / 1024
        d["issue_bound_us_at_2p4GHz"] = d["valu_issue_cycles_per_simd"] / 2400.0
        d["issue_utilisation_if_2p4GHz"] = d["issue_bound_us_at_2p4GHz"] / (d["mean_ns_profiled"] * 1e-3)
    res.append(d)
json.dump(res, open(os.path.join(out, tag + "_long_pmc.json"), "w"), indent=1)
for d in res:
    print(json.dumps(d))
PY
rm -rf $OUT/${TAG}_long_pmc   # (raw traces: gpurun_out/ is capped at 64 MiB)

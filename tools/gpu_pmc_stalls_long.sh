#!/bin/bash
# Where a lone wave's cycles go in the LONG-CHAIN kernels (rocprofv3 --pmc, its own runs, kernel trace only): issue / wait
# breakdown of N = 12 / K = 32, N = 10 / K = 32, N = 10 / K = 16 at B = 100k -> gpurun_out/${TAG}_long_stalls.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
for shape in "12 32" "10 32" "10 16" "8 32"; do
  set -- $shape
  t=${TAG}_n$1k$2
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/${t}_pmc -o p -- python $R/tools/long_chain_driver.py $1 $2 100000 20 > /dev/null 2> $OUT/${t}_pmc.err
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/${t}_pmc2 -o p -- python $R/tools/long_chain_driver.py $1 $2 100000 20 > /dev/null 2> $OUT/${t}_pmc2.err
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $OUT/${t}_pmc3 -o p -- python $R/tools/long_chain_driver.py $1 $2 100000 20 > /dev/null 2> $OUT/${t}_pmc3.err
done
python - $OUT $TAG <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
allres = {}
for d in sorted(glob.glob(os.path.join(out, tag + "_n*k*_pmc"))):
    base = d[:-4]
    res = {}
    for sub in ("_pmc", "_pmc2", "_pmc3"):
        f = glob.glob(os.path.join(base + sub, "**", "*counter_collection.csv"), recursive=True)
        if not f: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "mtg_solve" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                res["kernel"] = r["Kernel_Name"][:110]
        for name, v in acc.items():
            res[name] = sum(v[len(v) // 2:]) / len(v[len(v) // 2:])
        tr = glob.glob(os.path.join(base + sub, "**", "*kernel_trace.csv"), recursive=True)
        if tr and sub == "_pmc":
            du = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0])) if "mtg_solve" in r["Kernel_Name"]]
            res["mean_us_profiled"] = sum(du[len(du) // 2:]) / len(du[len(du) // 2:]) * 1e-3
    wc = res.get("SQ_WAVE_CYCLES") or 1
    res["fractions_of_SQ_WAVE_CYCLES"] = {n: round(v / wc, 4) for n, v in res.items() if isinstance(v, float) and (n.startswith("SQ_WAIT") or n.startswith("SQ_ACTIVE") or n.startswith("SQ_INST_CYCLES"))}
    if "SQ_INSTS_VALU" in res and "mean_us_profiled" in res:
        res["fp64_issue_utilisation_at_2p4GHz"] = res["SQ_INSTS_VALU"] * 4 / 1024 / 2400.0 / res["mean_us_profiled"]
    allres[os.path.basename(base)] = res
    print(os.path.basename(base), json.dumps(res))
json.dump(allres, open(os.path.join(out, tag + "_long_stalls.json"), "w"), indent=1)
PY
rm -rf $OUT/${TAG}_n*k*_pmc $OUT/${TAG}_n*k*_pmc2 $OUT/${TAG}_n*k*_pmc3

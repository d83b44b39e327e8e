#!/bin/bash
# Where a wave's cycles go (rocprofv3 --pmc, its own run): issue / wait breakdown of the solve kernels at B = 125k.
# usage: gpu_pmc_stalls.sh TAG [env assignments and bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/${TAG}_pmc -o p -- python $R/bench.py --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OUT/${TAG}_pmc.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/${TAG}_pmc2 -o p -- python $R/bench.py --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OUT/${TAG}_pmc2.err
python - $OUT $TAG <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for sub in ("_pmc", "_pmc2"):
    f = glob.glob(os.path.join(out, tag + sub, "**", "*counter_collection.csv"), recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "mtg_solve" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        for name, v in c.items():
            res[k][name] = sum(v[len(v) // 2:]) / len(v[len(v) // 2:])     # later half of the launches (warm)
        res[k]["launches" + sub] = len(next(iter(c.values())))
json.dump(res, open(os.path.join(out, tag + "_stalls.json"), "w"), indent=1)
for k, d in res.items():
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(k)
    print("   ", {n: round(v / wc, 3) for n, v in d.items() if n.startswith("SQ_WAIT") or n.startswith("SQ_ACTIVE")}, "(fractions of SQ_WAVE_CYCLES)")
    print("   ", {n: v for n, v in d.items() if n.startswith("SQ_INSTS") or n in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INST_CYCLES_VMEM")})
PY

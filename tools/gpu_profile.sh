#!/bin/bash
# Profiles of the bench command: rocprofv3 kernel stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE) -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r01}
B=${2:-10000}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --batch $B"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o s -- $CMD > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o p -- $CMD > /dev/null 2> $OUT/${TAG}_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o p -- $CMD > /dev/null 2> $OUT/${TAG}_write.err
python - $OUT $TAG $B <<'PY'
import csv, sys, glob, json, os
out, tag, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
kernel = None
def pmc(d, name):
    f = glob.glob(os.path.join(out, f"{tag}_{d}", "**", "*counter_collection.csv"), recursive=True)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Kernel_Name"] == kernel and r["Counter_Name"] == name]
    return sum(vals) / len(vals), len(vals)
stats = glob.glob(os.path.join(out, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)[0]
# the bench workload's kernel = the mtg_solve instantiation with the most calls (bench.py also times one 125k launch)
row = max((r for r in csv.DictReader(open(stats)) if "mtg_solve" in r["Name"]), key=lambda r: int(r["Calls"]))
kernel = row["Name"]
fetch, n1 = pmc("fetch", "FETCH_SIZE")
write, n2 = pmc("write", "WRITE_SIZE")
# rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream
# (MI355X_MICROARCH.md, HBM section) -> doubled.  WRITE_SIZE is used as reported (uncalibrated per the guide).
res = {"batch": B, "kernel": row["Name"], "kernel_avg_ns": float(row["AverageNs"]), "calls": int(row["Calls"]),
       "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "hbm_read_bytes_per_launch": 2 * fetch * 1024, "hbm_write_bytes_per_launch": write * 1024,
       "hbm_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
       "algorithmic_bytes_per_launch": B * 2392,
       "note": "separate --pmc passes; FETCH_SIZE doubled per the gfx950 correction; units KiB"}
json.dump(res, open(os.path.join(out, f"{tag}_b{B}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res))
PY
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_b${B}_kernel_stats.csv

#!/bin/bash
# Round-2 profiles of the bench command: rocprofv3 kernel stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE), with
# the counters calibrated in the same visit on streams of known size (tools/micro/fetch_calib.hip) -> gpurun_out/
# usage: gpu_profile2.sh TAG [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r02}; shift
ARGS="$*"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-extras $ARGS"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o s -- $CMD > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o p -- $CMD > /dev/null 2> $OUT/${TAG}_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o p -- $CMD > /dev/null 2> $OUT/${TAG}_write.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_calf -o p -- $R/tools/micro/fetch_calib > /dev/null 2> $OUT/${TAG}_calf.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_calw -o p -- $R/tools/micro/fetch_calib > /dev/null 2> $OUT/${TAG}_calw.err
python - $OUT $TAG "$ARGS" <<'PY'
import csv, sys, glob, json, os
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
def counters(d):
    f = glob.glob(os.path.join(out, f"{tag}_{d}", "**", "*counter_collection.csv"), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
def mean(rows, kernel_sub, name):
    v = [float(r["Counter_Value"]) for r in rows if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == name]
    return (sum(v) / len(v), len(v)) if v else (None, 0)
GiB2 = float(2 << 30)
calf, calw = counters("calf"), counters("calw")
cal = {}
for k in ("k_read8", "k_read16"):
    v, _ = mean(calf, k, "FETCH_SIZE")
    cal[k] = None if v is None else GiB2 / (v * 1024)      # true bytes per counted KiB-byte
v, _ = mean(calw, "k_write16", "WRITE_SIZE")
cal["k_write16"] = None if v is None else GiB2 / (v * 1024)
stats = glob.glob(os.path.join(out, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(stats)) if "mtg_solve" in r["Name"]]
row = max(rows, key=lambda r: int(r["Calls"]))
kernel = row["Name"]
fetch, n1 = mean(counters("fetch"), kernel, "FETCH_SIZE")
write, n2 = mean(counters("write"), kernel, "WRITE_SIZE")
bench = json.loads([l for l in open(os.path.join(out, f"{tag}_stats_bench.json")) if l.startswith("{")][-1])
B = bench["roofline"]["bytes_per_launch"] // bench["config"]["bytes_per_trajectory"]
f8 = cal.get("k_read8") or 1.0
fw = cal.get("k_write16") or 1.0
res = {"batch": B, "bench_args": args, "kernel": kernel, "kernel_avg_ns": float(row["AverageNs"]), "calls": int(row["Calls"]),
       "bench_kernel_us_hip_events": bench["roofline"]["kernel_us"], "buffer_sets": bench["config"]["buffer_sets"],
       "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "calibration_true_bytes_per_counted_byte": cal,
       "hbm_read_bytes_per_launch": None if fetch is None else f8 * fetch * 1024,
       "hbm_write_bytes_per_launch": None if write is None else fw * write * 1024,
       "hbm_bytes_per_launch": None if fetch is None or write is None else f8 * fetch * 1024 + fw * write * 1024,
       "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
       "note": "separate --pmc passes; counters in KiB, scaled by the factors measured in the same visit on 2 GiB streams "
               "(8 B/lane loads for FETCH_SIZE -- the kernels' input load shape --, 16 B/lane stores for WRITE_SIZE)"}
json.dump(res, open(os.path.join(out, f"{tag}_b{B}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res))
import shutil
shutil.copy(stats, os.path.join(out, f"{tag}_b{B}_kernel_stats.csv"))
PY

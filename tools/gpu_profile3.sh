#!/bin/bash
# Round-3 profiles of the bench command: rocprofv3 kernel trace + stats, and separate PMC passes (FETCH_SIZE, WRITE_SIZE)
# with the counters calibrated in the same visit on streams of known size (tools/micro/fetch_calib.hip) -> gpurun_out/.
# The timed region of the bench is ONE persistent launch over `steps` batches (mtg_solve_linear_sequence): the figures are
# taken from the LAST dispatch of the solve kernel in the trace (= the timed launch; the earlier ones are the buffer
# pre-touch and the warm-up), per launch and per batch.
# usage: gpu_profile3.sh TAG [bench args...]      (default bench args: the driver's --steps 20 --warmup 5)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r03}; shift
ARGS="${*:---steps 20 --warmup 5}"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --no-extras --no-parity $ARGS"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o s -- $CMD > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o p -- $CMD > /dev/null 2> $OUT/${TAG}_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o p -- $CMD > /dev/null 2> $OUT/${TAG}_write.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_calf -o p -- $R/tools/micro/fetch_calib > /dev/null 2> $OUT/${TAG}_calf.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_calw -o p -- $R/tools/micro/fetch_calib > /dev/null 2> $OUT/${TAG}_calw.err
python - $OUT $TAG "$ARGS" <<'PY'
import csv, sys, glob, json, os, shutil
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
def table(d, suffix):
    f = glob.glob(os.path.join(out, f"{tag}_{d}", "**", "*" + suffix), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
def mean(rows, kernel_sub, name):
    v = [float(r["Counter_Value"]) for r in rows if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == name]
    return (sum(v) / len(v)) if v else None
def last(rows, kernel, name):
    v = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in rows if r["Kernel_Name"] == kernel and r["Counter_Name"] == name]
    return max(v)[1] if v else None
GiB2 = float(2 << 30)
calf, calw = table("calf", "counter_collection.csv"), table("calw", "counter_collection.csv")
cal = {}
for k in ("k_read8", "k_read16"):
    v = mean(calf, k, "FETCH_SIZE")
    cal[k] = None if v is None else GiB2 / (v * 1024)      # true bytes per counted KiB-byte
v = mean(calw, "k_write16", "WRITE_SIZE")
cal["k_write16"] = None if v is None else GiB2 / (v * 1024)
bench = json.loads([l for l in open(os.path.join(out, f"{tag}_stats_bench.json")) if l.startswith("{")][-1])
trace = [r for r in table("stats", "kernel_trace.csv") if "mtg_solve" in r["Kernel_Name"]]
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
n_timed = bench["roofline"]["launches"]
timed = trace[-n_timed:]
kernel = timed[-1]["Kernel_Name"]
dur_ns = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
span_ns = int(timed[-1]["End_Timestamp"]) - int(timed[0]["Start_Timestamp"])
steps = bench["steps"]
fetch = last(table("fetch", "counter_collection.csv"), kernel, "FETCH_SIZE")
write = last(table("write", "counter_collection.csv"), kernel, "WRITE_SIZE")
f8 = cal.get("k_read8") or 1.0
fw = cal.get("k_write16") or 1.0
rf = bench["roofline"]
# steps of the LAST timed launch (a queue of more than 96 batches is cut: 200 steps = 96 + 96 + 8): its counters / its steps
per_launch_steps = steps - (rf["batches_per_launch"] or steps) * (n_timed - 1) if n_timed > 1 else steps
res = {"bench_args": args, "kernel": kernel, "timed_launches": n_timed, "batches_per_launch": rf["batches_per_launch"],
       "trajectories_per_step": bench["config"]["trajectories_per_step"],
       "rocprof_timed_launch_ns": dur_ns, "rocprof_us_per_step": span_ns * 1e-3 / steps,
       "bench_kernel_us_hip_events": rf["kernel_us"], "bench_device_us_per_step": rf["device_us_per_step"],
       "buffer_sets": bench["config"]["buffer_sets"],
       "FETCH_SIZE_KiB_raw_last_launch": fetch, "WRITE_SIZE_KiB_raw_last_launch": write,
       "calibration_true_bytes_per_counted_byte": cal,
       "hbm_read_bytes_per_step": None if fetch is None else f8 * fetch * 1024 / per_launch_steps,
       "hbm_write_bytes_per_step": None if write is None else fw * write * 1024 / per_launch_steps,
       "hbm_bytes_per_step": None if fetch is None or write is None else (f8 * fetch + fw * write) * 1024 / per_launch_steps,
       "algorithmic_bytes_per_step": rf["bytes_per_step"],
       "batch": bench["config"]["trajectories_per_step"],
       "hbm_bytes_per_launch": None if fetch is None or write is None else (f8 * fetch + fw * write) * 1024,
       "note": "separate --pmc passes; counters in KiB, scaled by the factors measured in the same visit on 2 GiB streams "
               "(8 B/lane loads for FETCH_SIZE -- the kernels' input load shape --, 16 B/lane stores for WRITE_SIZE); "
               "counters and durations of the LAST launch of the solve kernel = the bench's timed launch"}
json.dump(res, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res))
st = glob.glob(os.path.join(out, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
with open(os.path.join(out, f"{tag}_kernel_trace_solve_launches.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Duration_ns", "Grid_Size", "Workgroup_Size", "LDS", "VGPR", "SGPR"])
    for r in trace:
        w.writerow([r["Kernel_Name"], r["Start_Timestamp"], r["End_Timestamp"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                    r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")),
                    r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("SGPR_Count", "")])
PY
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_calf $OUT/${TAG}_calw   # (raw traces: gpurun_out/ is capped at 64 MiB)

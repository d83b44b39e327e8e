#!/bin/bash
# Is the slower first timed region of a fresh process a CLOCK effect?  GRBM_GUI_ACTIVE (GPU-busy cycles) per dispatch next to the
# dispatch duration from the kernel trace: cycles / ns = the clock the launch ran at.  The bench process measures the cold region,
# settles for 50 ms, then measures the timed region: the two 20-batch launches are the ones compared.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
sleep 15
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/clk -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/clk.err
python - $OUT > $OUT/r03x_clock_ramp.txt <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(os.path.join(out, "clk", "**", "*kernel_trace.csv"), recursive=True)[0]))}
rows = []
for r in csv.DictReader(open(glob.glob(os.path.join(out, "clk", "**", "*counter_collection.csv"), recursive=True)[0])):
    if "mtg_solve" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        t = tr[r["Dispatch_Id"]]
        ns = int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
        rows.append((int(t["Start_Timestamp"]), ns, float(r["Counter_Value"])))
rows.sort()
t0 = rows[0][0]
print("solve launches:", len(rows), "(GRBM_GUI_ACTIVE is summed over the 8 XCDs: GHz = cycles / ns / 8)")
def line(i):
    ts, ns, cyc = rows[i]
    return "  #%3d at %+9.2f ms: %7.1f us  %.3f GHz" % (i, (ts - t0) * 1e-6, ns * 1e-3, cyc / ns / 8)
# order of the process: 16 pre-touch batches (one launch), cold region = 5-batch warm-up + 20-batch launch, ~50 ms of 16-batch
# launches, 5-batch warm-up + the timed 20-batch launch
for i in list(range(0, min(8, len(rows)))) + list(range(len(rows) // 2, len(rows) // 2 + 3)) + list(range(max(0, len(rows) - 5), len(rows))):
    print(line(i))
PY
cat $OUT/r03x_clock_ramp.txt
rm -rf $OUT/clk

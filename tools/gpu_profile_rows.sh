#!/bin/bash
# VALU instructions per call of the compute-bound SURVEY 8(f) rows at the bench's size (rocprofv3 --pmc, its own runs, kernel
# trace only) -> gpurun_out/<TAG>_next_rows_pmc.json: {row: {valu_insts_per_call, kernels: {...}}}.  bench.py turns them into the
# rows' FP64-issue roofline (one VALU instruction per 4 cycles and SIMD = the 78.6 TFLOP/s FP64 vector peak for FMAs).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r04}
CALLS=4
cd /tmp && export TMPDIR=/tmp
for row in extrema time_scaling mellinger; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_rows_$row -o p -- python $R/tools/prof_next_rows.py $row $CALLS > /dev/null 2> $OUT/${TAG}_rows_$row.err
  # round 6: what the VALU instructions ARE (its own pass) -- the rows' FP64 fraction from the FP64 instructions, not from "every
  # VALU instruction is an FMA"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/${TAG}_rows_${row}_f64 -o p -- python $R/tools/prof_next_rows.py $row $CALLS > /dev/null 2> $OUT/${TAG}_rows_${row}_f64.err
done
python - $OUT $TAG $CALLS <<'PY'
import csv, sys, glob, os, collections, json, re
out, tag, calls = sys.argv[1], sys.argv[2], int(sys.argv[3])
res = {}
for row in ("extrema", "time_scaling", "mellinger"):
    f = glob.glob(os.path.join(out, f"{tag}_rows_{row}", "**", "*counter_collection.csv"), recursive=True)
    f += glob.glob(os.path.join(out, f"{tag}_rows_{row}_f64", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in [r_ for ff in f for r_ in csv.DictReader(open(ff))]:
        k = r["Kernel_Name"]
        if "mtg_" not in k or "mtg_solve_dl" in k or "slab" in k:     # (the set-up solve is not part of the row)
            continue
        name = re.search(r"mtg_\w+(<[^>]*>)?", k).group(0)[:60]
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
    res[row] = {"calls": calls, "valu_insts_per_call": sum(d.get("SQ_INSTS_VALU", 0.0) for d in per.values()) / calls,
                "waves_per_call": sum(d.get("SQ_WAVES", 0.0) for d in per.values()) / calls,
                "f64_insts_per_call": {c: sum(d.get("SQ_INSTS_VALU_" + c, 0.0) for d in per.values()) / calls for c in ("FMA_F64", "MUL_F64", "ADD_F64", "TRANS_F64")},
                "kernels": {k: {c: v / calls for c, v in d.items()} for k, d in per.items()},
                "what": "10k x 8-segment N = 10 trajectories (the bench's extra.next sizes); rocprofv3 --pmc, kernel trace only"}
json.dump(res, open(os.path.join(out, f"{tag}_next_rows_pmc.json"), "w"), indent=1)
print(json.dumps({k: {"valu_insts_per_call": v["valu_insts_per_call"], "waves_per_call": v["waves_per_call"]} for k, v in res.items()}))
PY
rm -rf $OUT/${TAG}_rows_extrema* $OUT/${TAG}_rows_time_scaling* $OUT/${TAG}_rows_mellinger*

#!/bin/bash
# rocprofv3 kernel stats + PMC passes for the sampler / extrema kernels -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/prof_next.py 100000 5"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_next_stats -o s -- $CMD > /dev/null 2> $OUT/${TAG}_next_stats.err
cp $(find $OUT/${TAG}_next_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_next_kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/${TAG}_next_$n -o p -- $CMD > /dev/null 2> $OUT/${TAG}_next_$n.err
done
python - $OUT $TAG <<'PY'
import csv, sys, glob, os, collections, json, re
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, f"{tag}_next_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mtg_" in k:
            acc[re.search(r"mtg_\w+(<[\d, ]+>)?", k).group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(os.path.join(out, f"{tag}_next_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
head -12 $OUT/${TAG}_next_kernel_stats.csv | cut -c1-160
rm -rf $OUT/${TAG}_next_stats $OUT/${TAG}_next_FETCH_SIZE $OUT/${TAG}_next_WRITE_SIZE $OUT/${TAG}_next_SQ_INSTS_VALU $OUT/${TAG}_next_SQ_WAIT_INST_ANY   # (raw traces: gpurun_out/ is capped at 64 MiB)

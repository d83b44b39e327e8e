#!/bin/bash
# Round 6, visit b: new tests (N2 against the reference's own member, the basic-solution flag through the batched entries, the
# null-space relation), the bench line with `sustained` / `value_aos_inputs`, the mtg_comm gather as the default.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06b; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -n 4 > $OUT/pytest_gpu.txt 2>&1; tail -n 30 $OUT/pytest_gpu.txt | cut -c1-400
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err
timeout 300 python bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/bench2.err | grep "^{" > $OUT/bench_two_ranks_one_gpu.json
timeout 300 python bench.py --gpus 1 --exercise-collectives --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>$OUT/bench3.err | grep "^{" > $OUT/bench_one_rank_rccl.json
timeout 300 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config5.json 2>> $OUT/bench.err
for f in bench_driver_args bench_two_ranks_one_gpu bench_one_rank_rccl bench_config5; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "value %.4g" % d["value"], "frac %.3f" % d["roofline"]["frac"], "cold", d.get("value_cold"))
    print("  sustained", json.dumps(d.get("sustained"))[:600])
    print("  aos", json.dumps(d.get("value_aos_inputs"))[:500])
    print("  other", json.dumps({k: (d.get("value_other_form") or {}).get(k) for k in ("value", "roofline_frac")}))
    print("  gather", json.dumps(d.get("gather"))[:900])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -c 600 $OUT/bench2.err $OUT/bench3.err

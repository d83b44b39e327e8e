#!/bin/bash
# Round 4, visit J: host cost of the timed library call, this tree vs the round-3 library (A/B in separate processes, interleaved);
# the RCCL one-rank tests; config 3 with plain and padded SoA rows.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04j; mkdir -p $OUT; cd $R
for i in 1 2 3; do
  python tools/enqueue_cost.py 2>/dev/null | grep "^{" >> $OUT/enqueue_cost.jsonl
  python tools/enqueue_cost.py _r3pkg 2>/dev/null | grep "^{" >> $OUT/enqueue_cost.jsonl
done
cat $OUT/enqueue_cost.jsonl | cut -c1-400
timeout 900 python -m pytest tests/test_bench.py -m gpu -q -x > $OUT/pytest_bench.txt 2>&1; tail -n 5 $OUT/pytest_bench.txt | cut -c1-400
for i in 1 2; do
  for lay in soa soa16; do
    python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --layout $lay 2>/dev/null | grep "^{" > $OUT/bench_config3_${lay}_$i.json
    python - $OUT/bench_config3_${lay}_$i.json $lay <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
print("config 3", sys.argv[2], "dev us/step %.2f frac %.4f value %.4g parity %s" % (r["device_us_per_step"], r["frac"], d["value"], d["parity"]["ok"]))
PY
  done
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" > $OUT/bench_driver_args_short.json
python - $OUT/bench_driver_args_short.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); print("config 2", d["value"], d["roofline"]["frac"], d["timed_region_wall_us"])
PY

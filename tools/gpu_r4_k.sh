#!/bin/bash
# Round 4, visit K: where the timed region's host time goes -- repeats of the bench's own timed region, and the library call
# in a loop (this tree vs the round-3 library).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04k; mkdir -p $OUT; cd $R
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --timed-repeats 6 2>/dev/null | grep "^{" > $OUT/bench_repeats_$i.json
  python - $OUT/bench_repeats_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = lambda b: {k: round(v, 1) for k, v in b.items()}
print("cold  ", r(d["cold_start"]["wall_breakdown_us"]))
print("timed ", r(d["timed_region_wall_us"]), "value %.4g frac %.3f" % (d["value"], d["roofline"]["frac"]))
for b in d["extra"]["timed_region_repeats"]: print("repeat", r(b))
PY
done
for i in 1 2; do
  python tools/enqueue_cost.py 2>/dev/null | grep "^{" >> $OUT/enqueue_cost.jsonl
  python tools/enqueue_cost.py _r3pkg 2>/dev/null | grep "^{" >> $OUT/enqueue_cost.jsonl
done
cut -c1-700 $OUT/enqueue_cost.jsonl

#!/bin/bash
# Round 4, visit L: the bench with the settle phase over the warm-up's buffer sets (prefill before it): host cost of the timed call.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04l; mkdir -p $OUT; cd $R
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --timed-repeats 2 2>$OUT/err_$i.txt | grep "^{" > $OUT/bench_repeats_$i.json
  python - $OUT/bench_repeats_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = lambda b: {k: round(v, 1) for k, v in b.items()}
print("cold  ", r(d["cold_start"]["wall_breakdown_us"]))
print("timed ", r(d["timed_region_wall_us"]), "value %.4g frac %.3f parity %s %s" % (d["value"], d["roofline"]["frac"], d["parity"]["ok"], d["parity"].get("prefilled")))
for b in d["extra"]["timed_region_repeats"]: print("repeat", r(b))
PY
done
for c in 3 5; do
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>>$OUT/err_c.txt | grep "^{" > $OUT/bench_config$c.json
  python - $OUT/bench_config$c.json $c <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("config", sys.argv[2], {k: round(v, 1) for k, v in d["timed_region_wall_us"].items()}, "value %.4g frac %.3f parity %s" % (d["value"], d["roofline"]["frac"], d["parity"]["ok"]))
PY
done
python bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-extras 2>>$OUT/err_c.txt | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('warmup 0: value %.4g parity %s' % (d['value'], d['parity']['ok']))"
timeout 900 python -m pytest tests/test_bench.py -m gpu -q -x > $OUT/pytest_bench.txt 2>&1; tail -n 3 $OUT/pytest_bench.txt | cut -c1-400

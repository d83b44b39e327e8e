#!/bin/bash
# Round 4, visit E: the whole GPU suite on the current library, cooperative form after the latency tweaks, bench lines.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04e; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -n 6 > $OUT/pytest_gpu.txt 2>&1; tail -n 8 $OUT/pytest_gpu.txt | cut -c1-600
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
BATCHES=64,1024,2048 timeout 300 python tools/bench_coop.py > $OUT/coop_vs_default.jsonl 2> $OUT/coop.err; python - $OUT/coop_vs_default.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if d["K"] >= 16: print(d["N"], d["K"], d["B"], d.get("auto_form"), d.get("auto_us"), d.get("coop_us"), d.get("coop_over_auto"))
PY
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err
python - $OUT/bench_driver_args.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value %.4g cold %.4g frac %.3f" % (d["value"], d.get("value_cold") or 0, r["frac"]), "parity", d["parity"]["ok"], d["parity"]["max_rel_err_vs_reference_build"])
print("next", {k: (round(v["us"], 1), (v.get("roofline") or {}).get("frac")) for k, v in d["extra"]["next"].items()})
PY

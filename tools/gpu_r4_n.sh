#!/bin/bash
# Round 4, visit N: factor store (LDL^T factor instead of G in the long chains' step storage): parity tests, long-chain timings, config 4.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04n; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -n 6 > $OUT/pytest_gpu.txt 2>&1; tail -n 6 $OUT/pytest_gpu.txt | cut -c1-300
python tools/bench_configs.py long 2>&1 | grep "^{" > $OUT/configs_long.jsonl
python - $OUT/configs_long.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["config"], "N", d["N"], "K", d["K"], "kernel_us %.1f frac %.3f" % (d["kernel_us"], d["frac_8TBps"]))
PY
(for n in 10 12 8; do KS=17,24,27,50,100 MAXKB=20000000 python tools/bench_other_k.py $n 2>&1 | grep "^{"; done) > $OUT/other_chain_lengths.jsonl
python - $OUT/other_chain_lengths.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k in ("N", "K", "B", "kernel_us", "frac_8TBps", "form", "us")})
PY
python bench.py --config 4 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_config4.json
python - $OUT/bench_config4.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
print("config 4 merged: dev us/step %.2f frac %.3f value %.4g parity %s; one launch per request: %s" % (r["device_us_per_step"], r["frac"], d["value"], d["parity"]["ok"], d["extra"]["one_launch_per_request"]["us_per_step"]))
print({k: (v["n"], v.get("max_rel_err_vs_port")) for k, v in d["parity"]["per_n"].items()})
PY
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
python - $OUT/bench_driver_args.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
print("config 2: value %.4g frac %.3f parity %s wall %s" % (d["value"], r["frac"], d["parity"]["ok"], {k: round(v, 1) for k, v in d["timed_region_wall_us"].items()}))
PY

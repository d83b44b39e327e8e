#!/bin/bash
# Round 3, second visit: full GPU suite; register-shared long chains (MtgCfg::kRegShared); balanced persistent grids.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03b; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt
python tools/bench_configs.py long 2>&1 | grep "^{" > $OUT/configs_long.jsonl
python tools/bench_configs.py 2>&1 | grep "^{" > $OUT/configs.jsonl
python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_config4.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_args.json 2>> $OUT/bench.err
MTG_NO_BALANCE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_args_nobalance.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver_args_b.json 2>> $OUT/bench.err
MTG_NO_BALANCE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver_args_nobalance_b.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --next > $OUT/bench_next.json 2>> $OUT/bench.err
python - $OUT <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in ("configs_long.jsonl", "configs.jsonl"):
    for l in open(os.path.join(out, f)):
        d = json.loads(l)
        print(f"{d['config']:14s} N={d['N']:2d} K={d['K']:2d} D={d['D']} B={d['B']:6d} {d['kernel_us']:8.2f} us  frac {d['frac_8TBps']:.3f}")
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d["roofline"]
        print(os.path.basename(f), "value %.3g" % d["value"], "dev us/step %.2f" % r["device_us_per_step"], "frac %.3f" % r["frac"],
              {k: round(v.get("us_per_step", v.get("kernel_us", 0)), 2) for k, v in d.get("extra", {}).items() if isinstance(v, dict) and ("us_per_step" in v or "kernel_us" in v)})
        if "next" in d.get("extra", {}):
            print(json.dumps(d["extra"]["next"]))
    except Exception as e:
        print(f, "FAILED", e)
PY

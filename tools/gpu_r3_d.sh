#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03d; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt
tests/cpp/test_veneer > $OUT/test_veneer.txt 2>&1; tail -n 3 $OUT/test_veneer.txt
python tools/bench_configs.py long 2>&1 | grep "^{" > $OUT/configs_long.jsonl
(for n in 10 8 12; do python tools/bench_other_k.py $n 2>&1 | grep "^{"; done) > $OUT/other_chain_lengths.jsonl
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
bash tools/gpu_pmc_long.sh > $OUT/long_pmc.txt 2>&1
cp $R/gpurun_out/r03_long_pmc.json $OUT/ 2>/dev/null
python - $OUT <<'PY'
import json, sys, os
out = sys.argv[1]
for l in open(os.path.join(out, "configs_long.jsonl")):
    d = json.loads(l); print(f"long N={d['N']:2d} K={d['K']:2d} B={d['B']:6d} {d['kernel_us']:8.2f} us  frac {d['frac_8TBps']:.3f}")
rows = [json.loads(l) for l in open(os.path.join(out, "other_chain_lengths.jsonl"))]
for n in (8, 10, 12):
    print("N=%d" % n, " ".join(f"K{r['K']}/{r['B']//1000}k:{r['kernel_us']}({r['frac_8TBps']})" for r in rows if r["N"] == n))
d = json.loads([l for l in open(os.path.join(out, "bench_driver_args.json")) if l.startswith("{")][-1])
print("driver args: frac %.3f dev us/step %.2f" % (d["roofline"]["frac"], d["roofline"]["device_us_per_step"]), json.dumps(d["extra"].get("next", {}))[:600])
PY
cat $OUT/long_pmc.txt | tail -n 8

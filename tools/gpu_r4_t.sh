#!/bin/bash
# Round 4, visit T: N = 10 / K = 32 without global workspace (10 + 6 steps), N = 12 / K = 32 with 2 + 4 + 10: affected tests, timings.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04t; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_dimlane.py tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -m gpu -q -n 6 -k "shape12 or shape13 or shape15 or merged or mixed or bucket or config4" > $OUT/pytest_subset.txt 2>&1; tail -n 5 $OUT/pytest_subset.txt | cut -c1-300
python tools/bench_configs.py long 2>&1 | grep "^{" > $OUT/configs_long.jsonl
python - $OUT/configs_long.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["config"], "N", d["N"], "K", d["K"], "kernel_us %.1f frac %.3f" % (d["kernel_us"], d["frac_8TBps"]))
PY
python bench.py --config 4 --steps 20 --warmup 5 --no-extras 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('config 4 merged: dev us/step %.2f frac %.3f parity %s'%(r['device_us_per_step'], r['frac'], d['parity']['ok']))"

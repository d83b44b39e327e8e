#!/bin/bash
# Round 4, visit R: factor store with partial elimination in the forward sweep: form-comparison tests, long-chain timings.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04r; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_dimlane.py tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -m gpu -q -n 6 > $OUT/pytest_subset.txt 2>&1; tail -n 6 $OUT/pytest_subset.txt | cut -c1-300
python tools/bench_configs.py long 2>&1 | grep "^{" > $OUT/configs_long.jsonl
python - $OUT/configs_long.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["config"], "N", d["N"], "K", d["K"], "kernel_us %.1f frac %.3f" % (d["kernel_us"], d["frac_8TBps"]))
PY
(for n in 10 12 8; do KS=24,50,100 MAXKB=20000000 python tools/bench_other_k.py $n 2>&1 | grep "^{"; done) > $OUT/other_chain_lengths.jsonl
python - $OUT/other_chain_lengths.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if d["B"] == 100000: print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k in ("N", "K", "B", "kernel_us", "frac_8TBps")})
PY
python bench.py --config 4 --steps 20 --warmup 5 --no-extras 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('config 4 merged: dev us/step %.2f frac %.3f parity %s'%(r['device_us_per_step'], r['frac'], d['parity']['ok']))"

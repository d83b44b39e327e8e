#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03u; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_bench.py -m gpu -x -q 2>&1 | tail -3
sleep 20
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
sleep 20
python bench.py --config 4 --steps 20 --warmup 5 --no-extras > $OUT/bench_config4.json 2>> $OUT/bench.err
python bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 5 --no-extras 2>/dev/null | grep "^{" > $OUT/bench_two_ranks_one_gpu.json
for f in bench_driver_args bench_config4 bench_two_ranks_one_gpu; do python - $OUT/$f.json <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split('/')[-1], 'value %.3g'%r['value'], 'frac %.3f'%r['roofline']['frac'], 'kernel_us %.1f'%r['roofline']['kernel_us'], 'cold', {k:(round(v,3) if isinstance(v,float) else v) for k,v in (r.get('cold_start') or {}).items() if k!='is'}, r['timed_region_wall_us'])
PY
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03c; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_parity.py -m gpu -x -q -k "bench_kernels or cross_structure or queue_and or mixed or merged" > $OUT/pytest.txt 2>&1
tail -n 3 $OUT/pytest.txt
for s in lpt rr lpt rr; do
  MTG_DL_ANY_SCHED=$s python bench.py --config 4 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config4 sched=$s dev us/step %.2f frac %.3f  200 steps: %.2f' % (d['roofline']['device_us_per_step'], d['roofline']['frac'], d['extra']['rotating_buffers_200_steps']['us_per_step']))"
done
python tools/bench_mixed.py 2500 merged 2>&1 | grep "^{" > $OUT/mixed_config4.jsonl
python tools/bench_mixed.py 10000 merged 2>&1 | grep "^{" >> $OUT/mixed_config4.jsonl
cat $OUT/mixed_config4.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['per_bucket'], d['us_per_mixed_batch'], round(d['frac_8TBps'],3))"

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03g; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_dimlane.py tests/test_gpu_sequence.py -m gpu -x -q -k "10-8-3 or 12-8-3 or 10-8-4 or 10-8-1 or status" > $OUT/pytest.txt 2>&1; tail -n 2 $OUT/pytest.txt
run() {
  python bench.py --no-cpu-baseline --no-next "${@:2}" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d.get('extra',{})
print('$1', 'dev us/step %.2f frac %.3f' % (d['roofline']['device_us_per_step'], d['roofline']['frac']), {k: round(v.get('us_per_step', v.get('kernel_us', 0)), 2) for k, v in e.items() if isinstance(v, dict)})"
}
export MTG_DL_OCC2=0
run "prefetch dimlane queue 20x10k" --steps 20 --warmup 5 --dims dimlane --no-extras
run "prefetch dimlane queue 96x10k" --steps 96 --warmup 96 --dims dimlane --no-extras
run "prefetch dimlane launches 10k" --steps 200 --warmup 20 --dims dimlane --sequence launches --no-extras
run "prefetch dimlane 125k launches" --steps 50 --warmup 10 --dims dimlane --batch 125000 --sequence launches --no-extras
run "prefetch dimlane 40k launches" --steps 50 --warmup 10 --dims dimlane --batch 40000 --sequence launches --no-extras
unset MTG_DL_OCC2
python tools/bench_configs.py 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['B']>=100000 or d['config'].startswith('config2'): print(f\"{d['config']:14s} N={d['N']:2d} K={d['K']:2d} D={d['D']} B={d['B']:6d} {d['kernel_us']:8.2f} us  frac {d['frac_8TBps']:.3f}\")"

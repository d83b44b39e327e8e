#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03e; mkdir -p $OUT
cd $R
tests/cpp/test_veneer > $OUT/test_veneer.txt 2>&1; tail -n 2 $OUT/test_veneer.txt
timeout 600 python -m pytest tests/test_gpu_dimlane.py tests/test_gpu_sequence.py -m gpu -x -q -k "10-8-3-4-1 or cross_over or status" > $OUT/pytest.txt 2>&1; tail -n 2 $OUT/pytest.txt
run() {  # label env... -- bench args
  python bench.py --no-cpu-baseline --no-next "${@:2}" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d.get('extra',{})
print('$1', 'dev us/step %.2f frac %.3f' % (d['roofline']['device_us_per_step'], d['roofline']['frac']), {k: round(v.get('us_per_step', v.get('kernel_us', 0)), 2) for k, v in e.items() if isinstance(v, dict)})"
}
for occ in 0 1; do
  export MTG_DL_OCC2=$occ
  run "occ2=$occ dimlane queue 20x10k" --steps 20 --warmup 5 --dims dimlane
  run "occ2=$occ dimlane queue 96x10k" --steps 96 --warmup 96 --dims dimlane --no-extras
  run "occ2=$occ dimlane launches 10k" --steps 200 --warmup 20 --dims dimlane --sequence launches --no-extras
  run "occ2=$occ dimlane 125k launches" --steps 50 --warmup 10 --dims dimlane --batch 125000 --sequence launches --no-extras
  run "occ2=$occ dimlane 40k launches" --steps 50 --warmup 10 --dims dimlane --batch 40000 --sequence launches --no-extras
done
unset MTG_DL_OCC2
run "default queue 20x10k" --steps 20 --warmup 5 --no-extras
run "default queue 96x10k" --steps 96 --warmup 96 --no-extras
run "default 125k launches" --steps 50 --warmup 10 --batch 125000 --sequence launches --no-extras

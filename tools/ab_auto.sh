#!/bin/bash
# default (auto) launch geometry across batch sizes
for B in "$@"; do
  python bench.py --no-cpu-baseline --batch $B --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=%s auto: %.2f us/step kernel %.2f us frac %.3f' % (sys.argv[1], d['ms_per_step']*1e3, d['roofline']['kernel_us'], d['roofline']['frac']))" $B
done

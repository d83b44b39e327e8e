#!/bin/bash
# split vs fused launch geometry of the specialised solve kernel across batch sizes (bench.py --dims)
for B in "$@"; do for dims in split fused; do
  python bench.py --no-cpu-baseline --batch $B --dims $dims --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=%s %s: kernel %.2f us frac %.3f' % (sys.argv[1], sys.argv[2], d['roofline']['kernel_us'], d['roofline']['frac']))" $B $dims
done; done

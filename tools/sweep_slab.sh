#!/bin/bash
# Slab-output fused kernel vs the older fused kernel (MTG_NO_SLAB=1) under the bench protocol.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for B in 30000 60000 100000 125000 250000; do
  for v in slab old; do
    if [ $v = old ]; then export MTG_NO_SLAB=1; else unset MTG_NO_SLAB; fi
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batch $B --dims fused --buffer-sets 8 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('N10K8 B=%6d %-5s rotating(8 sets) %8.2f us  resident %8.2f us' % ($B, '$v', d['roofline']['kernel_us'], d['extra']['resident_buffers']['launch_us']))"
  done
done

#!/bin/bash
# Launch-form cross-over under the bench protocol (rotating buffer sets): config 2 shape at several batch sizes, config 5.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for B in 5000 10000 20000 40000 80000 125000; do
  for d in dimlane fused split; do
    python bench.py --steps 300 --warmup 20 --no-cpu-baseline --batch $B --dims $d 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config2 B=%6d %-8s rotating %8.2f us  resident %8.2f us' % ($B, '$d', d['roofline']['kernel_us'], d['extra']['resident_buffers']['launch_us']))"
  done
done
for B in 12500 50000 100000; do
  for d in dimlane fused split; do
    python bench.py --config 5 --steps 200 --warmup 20 --no-cpu-baseline --batch $B --dims $d 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config5 B=%6d %-8s rotating %8.2f us  resident %8.2f us' % ($B, '$d', d['roofline']['kernel_us'], d['extra']['resident_buffers']['launch_us']))"
  done
done

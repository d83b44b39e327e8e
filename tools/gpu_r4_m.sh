#!/bin/bash
# Round 4, visit M: what makes the timed call's host time vary (8 ... 17 us)?  GC off / main thread pinned / both, 4 runs each, interleaved.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04m; mkdir -p $OUT; cd $R
for i in 1 2 3 4; do
  for v in "plain:" "gcoff:--lab-gc-off" "pin:--lab-pin" "both:--lab-gc-off --lab-pin"; do
    tag=${v%%:*}; fl=${v#*:}
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-next $fl 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['timed_region_wall_us']
print(json.dumps({'variant':'$tag','enqueue_call':round(b['enqueue_call'],1),'until_stop':round(b['until_stop_event'],1),'sync':round(b['barrier_and_synchronize'],1),'wall':round(b['wall'],1),'kernel_us':round(d['roofline']['kernel_us'],1),'value':d['value']}))" >> $OUT/variants.jsonl
  done
done
cat $OUT/variants.jsonl

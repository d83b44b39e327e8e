#!/bin/bash
mkdir -p gpurun_out/r06k
for rep in 1 2 3; do
for lib in old new; do
  if [ $lib = old ]; then export MTG_HIP_LIB=$PWD/tools/lab/bin/libmtg_hip_r06z.so; else unset MTG_HIP_LIB; fi
  python tools/bench_configs.py config5 2>&1 | grep config | sed "s/^/$lib /" | cut -c1-140
  python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib bench config5 frac', round(d['roofline']['frac'],4), 'dev us', round(d['roofline']['device_us_per_step'],2))"
done
done

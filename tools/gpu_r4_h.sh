#!/bin/bash
# Round 4, visit H: padded SoA layout (tests; config 5 bench with the plain and the padded stride; PMC traffic of the padded run).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04h; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_dimlane.py -m gpu -q -n 6 -k "padded or aos or cross_over" > $OUT/pytest.txt 2>&1; tail -n 4 $OUT/pytest.txt | cut -c1-400
for lay in soa soa16 soa soa16; do
  python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --layout $lay 2>> $OUT/bench.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$lay', 'value %.4g cold %.4g dev us/step %.2f frac %.3f' % (d['value'], d['value_cold'], r['device_us_per_step'], r['frac']), 'parity', d['parity']['ok'], d['parity']['max_rel_err_vs_reference_build'], 'one launch per step', round(d['value_other_form']['device_us_per_step'], 2))"
done
bash tools/gpu_profile3.sh r04_config5 --config 5 --steps 20 --warmup 5 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5 (default layout) rocprof us/step %.2f hbm/alg %.4f read %.4g write %.4g' % (d['rocprof_us_per_step'], d['hbm_bytes_per_step']/d['algorithmic_bytes_per_step'], d['hbm_read_bytes_per_step'], d['hbm_write_bytes_per_step']))"

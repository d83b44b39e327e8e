#!/bin/bash
# run-time-K body with the FULL-tail instantiations: parity (rt tests + fuzz) and the K > 32 sweep
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03w; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_dimlane.py tests/test_gpu_forms_fuzz.py -m gpu -x -q -n 6 -k "runtime_k or fuzz or every_form or entry_points or aos" > $OUT/pytest_rt.txt 2>&1; tail -3 $OUT/pytest_rt.txt
(for n in 10 8 12; do KS=40,50,64,100 MAXKB=20000000 python tools/bench_other_k.py $n 2>&1 | grep "^{"; done) > $OUT/long_k.jsonl; cat $OUT/long_k.jsonl
python - <<'PY'
import sys; sys.path.insert(0, '.')
import json, torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
for (n, k, dim, mi) in ((10, 24, 4, 7), (10, 32, 4, 7), (10, 40, 4, 1), (12, 24, 4, 1)):
    masks = m.ends_full_masks(n, k, mi)
    plan = m.Plan(ctx, n, dim, k, n // 2 - 1, masks)
    for B in (2500, 50000):
        with torch.cuda.stream(ctx.stream):
            t, f = m.random_waypoint_batch(B, k, dim, n, masks, seed=3, device="cuda", layout="soa")
            co = torch.empty((B, k, dim, n), dtype=torch.float64, device="cuda")
            plan.solve(t, f, layout="soa", coeffs=co); torch.cuda.synchronize(); ctx.sync()
            us = plan.time_last_solve(20)
        print(json.dumps(dict(N=n, K=k, D=dim, mi=mi, B=B, form=plan.launch_form(B), kernel_us=round(us, 2), frac=round(B * plan.bytes_per_trajectory / us * 1e-3 / 8000, 3))))
    plan.close()
PY

#!/bin/bash
# Round 4, visit I: extrema with FOUR brackets per refinement round (tests + timing), then the full GPU suite on the final code.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04i; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_extrema.py -m gpu -q > $OUT/pytest_extrema.txt 2>&1; tail -n 5 $OUT/pytest_extrema.txt | cut -c1-600
python - > $OUT/extrema_refine4.jsonl <<'PY'
import json, sys
sys.path.insert(0, ".")
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
masks = m.ends_full_masks(10, 8)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in (2500, 10000, 100000):
    with torch.cuda.stream(ctx.stream):
        t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=8, device="cuda")
        co, _, _ = plan.solve(t, f)
        torch.cuda.synchronize()
        for split in (-1, 1, 2, 5):
            ctx.set_option("extrema_split", split)
            def timed(fn, reps):
                fn(); torch.cuda.synchronize(); e0.record(ctx.stream)
                for _ in range(reps): fn()
                e1.record(ctx.stream); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / reps
            us_v = timed(lambda: m.minmax_magnitude(ctx, co, t, 1), 8)
            us_s = timed(lambda: m.scale_segment_times_to_meet_constraints(ctx, co.clone(), t.clone(), 2.0, 3.0), 3)
            print(json.dumps(dict(B=B, lanes_per_search=split, extrema_us=round(us_v, 1), time_scaling_us=round(us_s, 1))), flush=True)
PY
cat $OUT/extrema_refine4.jsonl
timeout 1500 python -m pytest tests -m gpu -q -n 6 > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench.err
python - $OUT/bench_driver_args.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]; p = d.get("parity") or {}
print("value %.3g" % d["value"], "frac %.3f" % r["frac"], "parity", p.get("ok"), "next", {k: (round(v["us"], 1), v.get("fp64_issue_frac")) for k, v in (d.get("extra", {}).get("next") or {}).items()})
PY

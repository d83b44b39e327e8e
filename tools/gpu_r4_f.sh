#!/bin/bash
# Round 4, visit F: extrema with shared root searches (tests + timing with one / two lanes per search at 10k and 100k).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04f; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_extrema.py -m gpu -q > $OUT/pytest_extrema.txt 2>&1; tail -n 5 $OUT/pytest_extrema.txt | cut -c1-600
python - > $OUT/extrema_split.jsonl <<'PY'
import json, sys
sys.path.insert(0, ".")
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
masks = m.ends_full_masks(10, 8)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in (2500, 10000, 30000, 100000):
    with torch.cuda.stream(ctx.stream):
        t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=8, device="cuda")
        co, _, _ = plan.solve(t, f)
        torch.cuda.synchronize()
        for split in (1, 2, 5, 6, -1):   # bits 0-1 lanes per search, bit 2 (4): per-level code bodies; -1 default (one body, by size)
            ctx.set_option("extrema_split", split)
            def timed(fn, reps):
                fn(); torch.cuda.synchronize(); e0.record(ctx.stream)
                for _ in range(reps): fn()
                e1.record(ctx.stream); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / reps
            us_v = timed(lambda: m.minmax_magnitude(ctx, co, t, 1), 5)
            us_s = timed(lambda: m.scale_segment_times_to_meet_constraints(ctx, co.clone(), t.clone(), 2.0, 3.0), 3)
            print(json.dumps(dict(B=B, lanes_per_search=split, extrema_us=round(us_v, 1), time_scaling_us=round(us_s, 1))), flush=True)
PY
cat $OUT/extrema_split.jsonl

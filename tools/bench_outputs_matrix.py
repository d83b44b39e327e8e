"""Coefficient-only solves vs solves that also return the cost / cost + d_P, per shape and batch: device time per launch over
rotating buffer sets (events on the library's stream), and the launch form taken by the coefficient-only call."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
nsets = 6
ctx = m.Context(0)
SHAPES = ((10, 8, 4, 3, 1), (10, 16, 4, 4, 7), (12, 16, 5, 3, 1), (10, 32, 4, 3, 1), (8, 16, 3, 3, 1))
if os.environ.get("SHAPES"):      # e.g. SHAPES="12,32,5,3,1;8,32,3,3,1"  (N, K, derivative, D, interior mask)
    SHAPES = tuple(tuple(int(x) for x in s_.split(",")) for s_ in os.environ["SHAPES"].split(";"))
BATCHES = tuple(int(x) for x in os.environ.get("BATCHES", "10000,100000").split(","))
for (N, K, d, D, mi) in SHAPES:
    masks = m.ends_full_masks(N, K, mi)
    plan = m.Plan(ctx, N, D, K, d, masks)
    for B in BATCHES:
        if K * B > 2_000_000:
            continue
        row = dict(N=N, K=K, D=D, B=B, form=plan.launch_form(B, "soa"))
        for mode in ("coeffs", "cost", "cost+free"):
            wc, wf = mode != "coeffs", mode == "cost+free"
            with torch.cuda.stream(ctx.stream):
                sets = []
                for s in range(nsets):
                    t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=5 + s, device="cuda", layout="soa")
                    sets.append((t, f, torch.empty((B, K, D, N), dtype=torch.float64, device="cuda"),
                                 torch.empty((D, plan.n_free, B), dtype=torch.float64, device="cuda") if wf else None,
                                 torch.empty((B,), dtype=torch.float64, device="cuda") if wc else None))
                def go(i):
                    t, f, co, fr, cost = sets[i % nsets]
                    plan.solve(t, f, layout="soa", coeffs=co, d_free=fr, cost=cost, want_free=wf, want_cost=wc)
                for i in range(nsets):
                    go(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 30
                e0.record(ctx.stream)
                for i in range(reps):
                    go(i)
                e1.record(ctx.stream)
                torch.cuda.synchronize()
            ctx.sync()
            row[mode + "_us"] = round(e0.elapsed_time(e1) * 1e3 / reps, 2)
        print(json.dumps(row), flush=True)
    plan.close()

"""AoS vs SoA device inputs (coefficient-only solves): device time per launch with rotating buffer sets, launch form taken.
usage: bench_layouts.py [sets = 8]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = m.Context(0)
for (N, K, d, D, mi) in ((10, 8, 4, 3, 1), (8, 8, 3, 3, 1), (10, 16, 4, 4, 7), (10, 32, 4, 3, 1)):
    masks = m.ends_full_masks(N, K, mi)
    plan = m.Plan(ctx, N, D, K, d, masks)
    for B in (10_000, 125_000):
        if K * B > 2_000_000:
            continue
        for lay in ("soa", "aos"):
            with torch.cuda.stream(ctx.stream):
                sets = []
                for s in range(nsets):
                    t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=5 + s, device="cuda", layout=lay)
                    sets.append((t, f, torch.empty((B, K, D, N), dtype=torch.float64, device="cuda")))
                for (t, f, co) in sets:
                    plan.solve(t, f, layout=lay, coeffs=co)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 40
                e0.record(ctx.stream)
                for i in range(reps):
                    t, f, co = sets[i % nsets]
                    plan.solve(t, f, layout=lay, coeffs=co)
                e1.record(ctx.stream)
                torch.cuda.synchronize()
            ctx.sync()
            us = e0.elapsed_time(e1) * 1e3 / reps
            print(json.dumps(dict(N=N, K=K, D=D, B=B, layout=lay, form=plan.launch_form(B, lay), us_per_launch=round(us, 2),
                                  frac_8TBps=round(B * plan.bytes_per_trajectory / us * 1e-3 / 8000.0, 3))), flush=True)
    plan.close()

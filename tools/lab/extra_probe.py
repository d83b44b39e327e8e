"""long chains: coefficient-only launch against the same launch with cost + d_P outputs (lab probe; B = 100k)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
B = 100_000
for (N, d) in ((8, 3), (10, 4), (12, 5)):
    for K in (8, 16, 32):
        masks = m.ends_full_masks(N, K, 1)
        plan = m.Plan(ctx, N, 3, K, d, masks)
        with torch.cuda.stream(ctx.stream):
            t, f = m.random_waypoint_batch(B, K, 3, N, masks, seed=11, device="cuda", layout="soa")
            co = torch.empty((B, K, 3, N), dtype=torch.float64, device="cuda")
            plan.solve(t, f, layout="soa", coeffs=co)
            torch.cuda.synchronize(); ctx.sync()
            us0 = plan.time_last_solve(10)
            plan.solve(t, f, layout="soa", coeffs=co, want_free=True, want_cost=True)
            torch.cuda.synchronize(); ctx.sync()
            us1 = plan.time_last_solve(10)
        print(json.dumps(dict(N=N, K=K, B=B, plain_us=round(us0, 1), with_cost_and_dP_us=round(us1, 1), ratio=round(us1 / us0, 2),
                              form=plan.launch_form(B, "soa", extra_outputs=True))))
        plan.close()

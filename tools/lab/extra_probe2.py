"""K = 32 with cost + d_P outputs: the dimension-in-lane extra-output twin against the rolled fused kernel (lab probe)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
for (N, d) in ((8, 3), (10, 4), (12, 5)):
    for K in (16, 32):
        for B in (2500, 10_000, 30_000, 100_000):
            masks = m.ends_full_masks(N, K, 1)
            plan = m.Plan(ctx, N, 3, K, d, masks)
            out = dict(N=N, K=K, B=B)
            for dims in ("dimlane", "fused"):
                with torch.cuda.stream(ctx.stream):
                    t, f = m.random_waypoint_batch(B, K, 3, N, masks, seed=11, device="cuda", layout="soa")
                    co = torch.empty((B, K, 3, N), dtype=torch.float64, device="cuda")
                    plan.solve(t, f, layout="soa", coeffs=co, want_free=True, want_cost=True, dims=dims)
                    torch.cuda.synchronize(); ctx.sync()
                    out[dims] = round(plan.time_last_solve(10), 1)
            print(json.dumps(out))
            plan.close()

"""Long chains at B = 100k: the default (dimension-in-lane) form against the fused form (one lane per trajectory-half, D right-hand
sides: the rolled run-time-K kernel for these shapes) and the dimension-split form."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
for n in (10, 12, 8):
    for K in (16, 32):
        masks = m.ends_full_masks(n, K, 1)
        plan = m.Plan(ctx, n, 3, K, n // 2 - 1, masks)
        B = 100_000
        with torch.cuda.stream(ctx.stream):
            t, f = m.random_waypoint_batch(B, K, 3, n, masks, seed=11, device="cuda", layout="soa")
            co = torch.empty((B, K, 3, n), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            out = dict(N=n, K=K, B=B)
            for dims in ("auto", "fused", "split"):
                plan.solve(t, f, layout="soa", coeffs=co, dims=dims)
                torch.cuda.synchronize(); ctx.sync()
                out[dims + "_us"] = round(min(plan.time_last_solve(10) for _ in range(2)), 1)
                out[dims + "_form"] = plan.kernel_variant
            print(json.dumps(out), flush=True)
        plan.close()

"""Row-cooperative form (launch form 7) against the default choice of the same plan: kernel time over (N, K, batch)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m

ctx = m.Context(0)
cases = [(n, k) for n in (10, 8, 12) for k in (8, 16, 32, 50, 100)]
batches = [int(x) for x in os.environ.get("BATCHES", "64,256,1024,2500,5000,10000,20000").split(",")]
for (n, k) in cases:
    d = n // 2 - 1
    masks = m.ends_full_masks(n, k, 1)
    plan = m.Plan(ctx, n, 3, k, d, masks)
    for B in batches:
        with torch.cuda.stream(ctx.stream):
            t, f = m.random_waypoint_batch(B, k, 3, n, masks, seed=11, device="cuda", layout="soa")
            co = torch.empty((B, k, 3, n), dtype=torch.float64, device="cuda")
            row = dict(N=n, K=k, B=B)
            for dims in ("auto", "coop"):
                if plan.launch_form(B, "soa", dims) != ("coop" if dims == "coop" else plan.launch_form(B, "soa", "auto")):
                    continue
                plan.solve(t, f, layout="soa", coeffs=co, dims=dims)
                torch.cuda.synchronize(); ctx.sync()
                us = plan.time_last_solve(20)
                row[dims + "_us"] = round(us, 2)
                row[dims + "_form"] = plan.launch_form(B, "soa", dims)
            if "coop_us" in row and "auto_us" in row:
                row["coop_over_auto"] = round(row["coop_us"] / row["auto_us"], 3)
            print(json.dumps(row), flush=True)
    plan.close()

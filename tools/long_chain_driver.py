"""A bare loop of one long-chain solve for profilers (rocprofv3 PC sampling / PMC): python tools/long_chain_driver.py N K B iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m

n, k, bsz, iters = (int(x) for x in (sys.argv[1:5] + ["12", "32", "100000", "30"][len(sys.argv) - 1:]))
ctx = m.Context(0)
masks = m.ends_full_masks(n, k, 1)
plan = m.Plan(ctx, n, 3, k, n // 2 - 1, masks)
with torch.cuda.stream(ctx.stream):
    t, f = m.random_waypoint_batch(bsz, k, 3, n, masks, seed=11, device="cuda", layout="soa")
    co = torch.empty((bsz, k, 3, n), dtype=torch.float64, device="cuda")
    for _ in range(iters):
        plan.solve(t, f, layout="soa", coeffs=co)
    torch.cuda.synchronize()
ctx.sync()
print("form", plan.launch_form(bsz, "soa"), "us", plan.time_last_solve(10))

"""Cost-only launches and the batched Mellinger step: kernel time vs the full solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
masks = m.ends_full_masks(10, 8)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream)
    for _ in range(n): fn()
    e1.record(ctx.stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
with torch.cuda.stream(ctx.stream):
    for B in (10_000, 125_000, 1_000_000):
        t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=8, device="cuda", layout="soa")
        co = torch.empty((B, 8, 3, 10), dtype=torch.float64, device="cuda"); cost = torch.empty((B,), dtype=torch.float64, device="cuda")
        us_full = timeit(lambda: plan.solve(t, f, layout="soa", coeffs=co))
        us_cost = timeit(lambda: plan.solve_cost_only(t, f, layout="soa", cost=cost))
        print(f"B={B}: full solve {us_full:.1f} us ({B/us_full:.0f} M traj/s) | cost-only {us_cost:.1f} us ({B/us_cost:.0f} M traj/s)")
    ta, fa = m.random_waypoint_batch(100_000, 8, 3, 10, masks, seed=9, device="cuda", layout="aos")
    us = timeit(lambda: m.mellinger_cost_and_gradient(plan, ta, fa), 5)
    print(f"Mellinger cost+gradient, 100k trajectories x 9 solves: {us:.0f} us per step = {100_000*9/us:.0f} M solves/s")

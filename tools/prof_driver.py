"""Runs the solve kernel a few times at a given batch size (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
layout = sys.argv[3] if len(sys.argv) > 3 else "soa"
masks = m.ends_full_masks(10, 8)
ctx = m.Context(0)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
with torch.cuda.stream(ctx.stream):
    t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=5, device="cuda", layout=layout)
    co = torch.empty((B, 8, 3, 10), dtype=torch.float64, device="cuda")
    for _ in range(reps):
        plan.solve(t, f, layout=layout, coeffs=co)
    torch.cuda.synchronize()
ctx.sync()
print("done", B, reps)

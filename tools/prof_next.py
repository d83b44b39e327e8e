"""Driver for profiling the next-step kernels (rows N3/N4): a few launches of the sampler and the extrema search."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = m.Context(0)
masks = m.ends_full_masks(10, 8)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
with torch.cuda.stream(ctx.stream):
    t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=8, device="cuda")
    co, _, _ = plan.solve(t, f)
    dt = float(t.sum(dim=1).max()) / 100
    ws = torch.empty((8 * B * 9,), dtype=torch.float64, device="cuda")
    for _ in range(reps):
        out = m.sample_range(ctx, co, t, 0.0, dt, 100, 5)
        seg, traj, idx = m.minmax_magnitude(ctx, co, t, 1)
        seg, traj, idx = m.minmax_magnitude(ctx, co, t, 2)
    ts, fs = t.t().contiguous(), f.permute(1, 2, 0).contiguous()
    for _ in range(reps):
        m.mellinger_cost_and_gradient(plan, ts, fs, layout="soa")       # row N2: (K + 1) x B cost-only solves in one launch
    c2, t2 = co.clone(), t.clone()
    m.scale_segment_times_to_meet_constraints(ctx, c2, t2, 2.0, 2.0, workspace=ws)
    torch.cuda.synchronize()

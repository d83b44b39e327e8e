"""One SURVEY 8(f) row at the bench's size (10k x 8-segment N = 10 trajectories), a few calls -- for rocprofv3 --pmc SQ_INSTS_VALU
(tools/gpu_profile_rows.sh): VALU instructions per call, the numerator of the rows' FP64-issue roofline in the bench line.
usage: prof_next_rows.py extrema|time_scaling|mellinger [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
row = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, K, D, N = 10_000, 8, 3, 10
ctx = m.Context(0)
masks = m.ends_full_masks(N, K)
plan = m.Plan(ctx, N, D, K, 4, masks)
with torch.cuda.stream(ctx.stream):
    t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=1234, device="cuda", layout="soa")
    co, _, _ = plan.solve(t, f, layout="soa")
    tt = t.t().contiguous()
    torch.cuda.synchronize()
    for _ in range(calls):
        if row == "extrema":
            m.minmax_magnitude(ctx, co, tt, 1)
        elif row == "time_scaling":
            m.scale_segment_times_to_meet_constraints(ctx, co.clone(), tt.clone(), 2.0, 3.0)
        elif row == "mellinger":
            m.mellinger_cost_and_gradient(plan, t, f, layout="soa")
    torch.cuda.synchronize()
print("calls", calls)

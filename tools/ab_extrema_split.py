"""Lanes per root search of the extrema kernels (measurement knob "extrema_split": 1, 2, 3 = four lanes in the whole launch;
0 = the default's two regions -- whole rounds of wavefronts at one lane per search, the surplus searches on two / four lanes
each), over batch sizes around the ones that matter: kernel time of mtg_minmax_magnitude (velocity, acceleration) and of
mtg_scale_segment_times_to_meet_constraints."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m

ctx = m.Context(0)
N, K, D = 10, 8, 3
masks = m.ends_full_masks(N, K)
plan = m.Plan(ctx, N, D, K, N // 2 - 1, masks)
with torch.cuda.stream(ctx.stream):
    for B in (2500, 5000, 8192, 10_000, 12_000, 16_384, 20_000, 30_000, 100_000):
        t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=8, device="cuda")
        co, _, _ = plan.solve(t, f)
        ws = torch.empty((8 * B * (K + 1),), dtype=torch.float64, device="cuda")
        row = {"B": B, "waves_one_lane_per_search": (B * K + 63) // 64}
        for opt, name in ((1, "one"), (2, "two"), (3, "four"), (0, "default")):
            ctx.set_option("extrema_split", opt)
            for der in (1, 2):
                m.minmax_magnitude(ctx, co, t, der); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ctx.stream)
                for _ in range(5):
                    m.minmax_magnitude(ctx, co, t, der)
                e1.record(ctx.stream); torch.cuda.synchronize()
                row[f"{name}_der{der}_us"] = round(e0.elapsed_time(e1) * 1e3 / 5, 1)
            best = 1e30
            for _ in range(3):
                c2, t2 = co.clone(), t.clone(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ctx.stream)
                m.scale_segment_times_to_meet_constraints(ctx, c2, t2, 2.0, 2.0, workspace=ws)
                e1.record(ctx.stream); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3)
            row[f"{name}_scale_us"] = round(best, 1)
        ctx.set_option("extrema_split", -1)
        print(json.dumps(row), flush=True)
plan.close()

"""Solves that also return cost and d_P (what every nlopt-style caller asks for): device time per launch with
rotating buffer sets.  usage: bench_extra_outputs.py [B = 125000] [sets = 8]; MTG_NO_SLAB_EXTRA=1 -> the older fused kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
B = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
nsets = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = m.Context(0)
for (N, K, d) in ((10, 8, 4), (8, 8, 3), (10, 4, 4)):
    masks = m.ends_full_masks(N, K, 1)
    plan = m.Plan(ctx, N, 3, K, d, masks)
    with torch.cuda.stream(ctx.stream):
        sets = []
        for s in range(nsets):
            t, f = m.random_waypoint_batch(B, K, 3, N, masks, seed=5 + s, device="cuda", layout="soa")
            sets.append((t, f, torch.empty((B, K, 3, N), dtype=torch.float64, device="cuda"),
                         torch.empty((3, plan.n_free, B), dtype=torch.float64, device="cuda"), torch.empty((B,), dtype=torch.float64, device="cuda")))
        for (t, f, co, fr, cost) in sets:
            plan.solve(t, f, layout="soa", coeffs=co, d_free=fr, cost=cost, want_free=True, want_cost=True, dims="fused")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        e0.record(ctx.stream)
        for i in range(reps):
            t, f, co, fr, cost = sets[i % nsets]
            plan.solve(t, f, layout="soa", coeffs=co, d_free=fr, cost=cost, want_free=True, want_cost=True, dims="fused")
        e1.record(ctx.stream)
        torch.cuda.synchronize()
    ctx.sync()
    us = e0.elapsed_time(e1) * 1e3 / reps
    by = plan.bytes_per_trajectory + 8 * (3 * plan.n_free + 1)
    print(json.dumps(dict(N=N, K=K, B=B, sets=nsets, older_kernel=bool(os.environ.get("MTG_NO_SLAB_EXTRA")), us_per_launch=round(us, 2),
                          frac_8TBps=round(B * by / us * 1e-3 / 8000.0, 3), bytes_per_traj=by)))
    plan.close()

"""Kernel times of the chain lengths between the BASELINE ones (usage: bench_other_k.py [N]; D = 3, standard masks): K = 3 .. 15 have
their own dimension-in-lane variants, K = 20 / 50 / 100 run the rolled (run-time-K) kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
NN = int(sys.argv[1]) if len(sys.argv) > 1 else 10
KS = [int(x) for x in os.environ["KS"].split(",")] if os.environ.get("KS") else (3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 17, 20, 24, 27, 31, 50, 100)
for K in KS:
    masks = m.ends_full_masks(NN, K, 1)
    plan = m.Plan(ctx, NN, 3, K, NN // 2 - 1, masks)
    for B in (2500, 100_000):
        if K * B > int(os.environ.get("MAXKB", 3_000_000)):
            continue
        with torch.cuda.stream(ctx.stream):
            t, f = m.random_waypoint_batch(B, K, 3, NN, masks, seed=11, device="cuda", layout="soa")
            co = torch.empty((B, K, 3, NN), dtype=torch.float64, device="cuda")
            plan.solve(t, f, layout="soa", coeffs=co)
            torch.cuda.synchronize(); ctx.sync()
            us = plan.time_last_solve(20)
        bpt = plan.bytes_per_trajectory
        print(json.dumps(dict(N=NN, K=K, B=B, variant=plan.kernel_variant, kernel_us=round(us, 2), frac_8TBps=round(B * bpt / us * 1e-3 / 8000.0, 3))))
    plan.close()

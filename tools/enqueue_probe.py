"""Host enqueue rate of mtg_solve_linear_sequence vs GPU execution rate (measurement only)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
import bench
masks = m.ends_full_masks(10, 8)
ctx = m.Context(0)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
with torch.cuda.stream(ctx.stream):
    sets = []
    for s in range(2):
        t, f = m.random_waypoint_batch(10000, 8, 3, 10, masks, seed=s, device="cuda", layout="soa")
        sets.append((t, f, torch.zeros((10000, 8, 3, 10), dtype=torch.float64, device="cuda")))
    loop = bench.SolveLoop(plan, sets[:1], "soa", "auto")
    loop.prepare(2000)
    loop.run(2000); torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter(); loop.run(2000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"enqueue {1e6*(t1-t0)/2000:.2f} us/launch, total {1e6*(t2-t0)/2000:.2f} us/launch")

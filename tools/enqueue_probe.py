"""Host enqueue rate of mtg_solve_linear_sequence vs GPU execution rate; sensitivity of the per-launch period to the
number of timed steps and to what ran before (measurement only)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
import bench
masks = m.ends_full_masks(10, 8)
ctx = m.Context(0)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 2
with torch.cuda.stream(ctx.stream):
    sets = []
    for s in range(nsets):
        t, f = m.random_waypoint_batch(10000, 8, 3, 10, masks, seed=s, device="cuda", layout="soa")
        sets.append((t, f, torch.zeros((10000, 8, 3, 10), dtype=torch.float64, device="cuda")))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, ss in (("resident", sets[:1]), ("rotating", sets)):
        loop = bench.SolveLoop(plan, ss, "soa", "auto")
        for steps in (20, 200, 2000, 200, 20):
            loop.prepare(steps)
            loop.run(steps); torch.cuda.synchronize()
            res = []
            for rep in range(3):
                t0 = time.perf_counter(); e0.record(ctx.stream); loop.run(steps); e1.record(ctx.stream); t1 = time.perf_counter()
                torch.cuda.synchronize(); t2 = time.perf_counter()
                res.append((1e6*(t1-t0)/steps, e0.elapsed_time(e1)*1e3/steps, 1e6*(t2-t0)/steps))
            print(name, "steps", steps, " ".join("enq %.2f ev %.2f wall %.2f |" % r for r in res))

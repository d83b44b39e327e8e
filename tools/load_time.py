"""Library load + first-launch cost (code-object load): wall time of import, context, first plan, first solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
import torch
torch.cuda.init()
t1 = time.perf_counter()
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
t2 = time.perf_counter()
masks = m.ends_full_masks(10, 8)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
t, f = m.random_waypoint_batch(1000, 8, 3, 10, masks, seed=1, device="cuda", layout="soa")
torch.cuda.synchronize()
t3 = time.perf_counter()
co, _, _ = plan.solve(t, f, layout="soa")
ctx.sync()
t4 = time.perf_counter()
co, _, _ = plan.solve(t, f, layout="soa")
ctx.sync()
t5 = time.perf_counter()
print(f"lib={os.path.basename(m._lib.LIB_PATH)} torch+init {t1-t0:.2f}s  import+context {t2-t1:.3f}s  plan+workload {t3-t2:.3f}s  first solve {t4-t3:.4f}s  second solve {(t5-t4)*1e6:.0f} us  finite={bool(torch.isfinite(co).all())}")

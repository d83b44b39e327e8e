"""B = 10k ablations (measurement only): kernel time with / without the coefficient stores, split / fused geometry."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m
from mav_trajectory_generation_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
masks = m.ends_full_masks(10, 8)
ctx = m.Context(0)
plan = m.Plan(ctx, 10, 3, 8, 4, masks)
with torch.cuda.stream(ctx.stream):
    t, f = m.random_waypoint_batch(B, 8, 3, 10, masks, seed=5, device="cuda", layout="soa")
    co = torch.empty((B, 8, 3, 10), dtype=torch.float64, device="cuda")
    cost = torch.empty((B,), dtype=torch.float64, device="cuda")
    lay = plan.layout(B, "soa")
    for name, fl in (("split", L.FLAG_SPLIT_DIMS), ("fused", L.FLAG_FUSED_DIMS)):
        plan.solve(t, f, layout="soa", coeffs=co, dims=name)
        torch.cuda.synchronize()
        a = plan.time_last_solve(300)
        rc = plan.lib.mtg_solve_linear(plan.handle, B, ctypes.byref(lay), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(f.data_ptr()),
                                       None, None, ctypes.c_void_p(cost.data_ptr()), L.FLAG_COST_ONLY | fl)
        assert rc == 0
        torch.cuda.synchronize()
        b = plan.time_last_solve(300)
        print(f"B={B} {name}: full {a:.2f} us   cost-only (no coefficient stores) {b:.2f} us")
ctx.sync()
